// spectral.h -- shared declarations of the spectral forward engine (spectral.hip; used by forward.hip).
#pragma once

struct SpecPlan; // tables of one spectral set (spectral.hip)

// One pass of one transform (a "segment" of a pass launch): out[j0 + n L] = DFT_radix( tw in[j + n len / radix] ).
struct SpecSeg {
	unsigned item0;           // first wave-item of the segment in its launch
	unsigned len;             // points of the transform
	unsigned L;               // product of the radices of the earlier passes
	unsigned radix;
	unsigned nfold;           // first inverse pass of a scale with fewer outputs than classes: entry q = sum of nfold rows q + p len
	unsigned tw_mul;          // twiddle e^{2 pi i n k / (L radix)} = table entry n k tw_mul
	unsigned last;            // 1: last pass of an inverse transform (stacks / coefficients)
	unsigned nvalid;          // last pass: outputs k < nvalid are coefficients (N_s = ceil(N / D) of the len points when the transform is longer than the trace)
	unsigned long long src, dst; // first row inside a trace block's region of the source / destination buffer
	unsigned long long coff;     // last pass: first coefficient of the scale
	double tau;                  // last pass: noise floor of the scale per unit of max |x|
};

// One accumulator of the multiply-and-fold kernel.
struct SpecSlot {
	unsigned ld;              // log2 D of the scale; 31: never completes inside the loop (partial sums of the classes, or an idle pad)
	unsigned lb;              // log2 (N_s / classes) of a scale that completes; 31 with ld == 31: idle pad slot
	unsigned long long goff;  // first row of the scale's folded spectrum in a trace block's region
};

// The decomposition of a many-trace batch that goes with a spectral set: trace-lane items for the finer octaves + scale table.
struct SpecDecomp {
	unsigned s_first = 0, s_end = 0; // the spectral set [s_first, s_end); scales from s_end on (filters too long for the transform window) stay on the direct kernel
	bool few = false, small = false; // few: no trace-lane table (rows in columns: spectral.hip's tspws_spectral_rows_*); small: built for < 8 trace blocks
	TlTable T;
	SpecPlan *sp = nullptr;
};
