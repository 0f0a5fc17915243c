// fwd_mfma_types.h -- work descriptors shared by the matrix-pipe forward kernels (fwd_mfma.h, fwd_mfma_spec.hip) and the host.
#pragma once

#ifndef FM_TAMAX
#define FM_TAMAX 8   /* tiles (of 4 outputs) per unit */
#endif
#define FM_KQCAP 8     /* tap steps (of 4 rows) per phase the kernels accept; longer filters stay on the VALU kernels */
#define FM_MAXGROUPS 64

// one work group of the matrix-pipe forward kernel: up to two voice pairs of an octave sharing one x image
struct FwdGroup {
	unsigned D, Ns, Mc, logMc, MC, cps, nsplit, css; // decimation, outputs, phases per chunk, chunks, chunks per split / per staged sub-split
	unsigned nob, upi, TQ, P, Pu, RT, NP;            // output blocks (4 TQ outputs), units per work item, tiles per unit, plane / unit pitch, rows staged, pairs
	unsigned bl_doubles, s0, bper;                   // LDS doubles reserved for the B tiles of a sub-split, first scale, B doubles per chunk
	long long cp;                                    // origin of the image rows
	unsigned Kq[2], rofs[2], nv[2];                  // per pair: tap steps, row offset into the image, voices
	unsigned long long bt_off[2];                    // per pair: B table offset (doubles)
	unsigned long long po[4];                        // per voice: offset of its [nsplit][Ns] partial block (double2)
};

struct FwdOffsets { unsigned off[FM_MAXGROUPS + 1]; }; // first work item of every group in one launch (groups left out have no items)

// fwd_mfma_spec.hip: kernels specialised on (TQ, Mc) and the tap steps of both pairs.  launch returns 1 when it launched.
int fwd_mfma_spec_launch(int is_float, unsigned tq, unsigned mc, unsigned kq0, unsigned kq1, unsigned items, size_t lds, void *stream, const void *x,
                         size_t ld, unsigned ntr, unsigned N, const FwdGroup *pd, unsigned ngroups, const FwdOffsets &offs, const double *bt, void *part,
                         size_t npart);
int fwd_mfma_spec_has(unsigned tq, unsigned mc, unsigned kq0, unsigned kq1);
