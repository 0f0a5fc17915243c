// resample.hip -- jackknife (two-stage), random subsampling, convergence curves.
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"
#include <atomic>
#include <chrono>
#include <string>
#include <unordered_map>

#define is_two_stage tspws_is_two_stage

// ------------------------------------------------------------------------------------------
// jackknife (TwoStage_jackknife_float, ts_pws1f_lib.c:719-831)
// ------------------------------------------------------------------------------------------
extern "C" int tspws_jackknife_plan(char *sel, const time_t *tm, size_t mtr, unsigned d, unsigned n, unsigned C)
{
	if (!sel || !tm) return 1;
	if (tm[0] == 0) return -2;
	std::vector<unsigned> bin(mtr), comb(d);
	for (size_t i = 0; i < mtr; i++) {
		struct tm g;
		gmtime_r(tm + i, &g);
		bin[i] = (unsigned)floor((double)(g.tm_yday * (int)n) / 365.); // day-of-year bin, :398-401
	}
	for (unsigned i = 0; i < d; i++) comb[i] = i;
	for (unsigned c = 0; c < C; c++) {
		if (c) { // lexicographic successor of the deleted-bin set, :405-414
			int i = (int)d - 1;
			while (i >= 0 && comb[i] >= n - d + (unsigned)i) i--;
			if (i < 0) break;
			comb[i]++;
			for (unsigned j = (unsigned)i + 1; j < d; j++) comb[j] = comb[j - 1] + 1;
		}
		char *row = sel + (size_t)c * mtr;
		for (size_t t = 0; t < mtr; t++) {
			row[t] = 1;
			for (unsigned i = 0; i < d; i++) if (bin[t] == comb[i]) row[t] = 0;
		}
	}
	return 0;
}

// replica linear stack in the time domain, :799-811: (sum_g P[g]) * (1/K)
// rows of replica c.  gps == 0: replica-major rows, P[(c Kmax + g) N]; gps > 0: the staged layout of the pipelined call,
// row(g, c) = g0 W + c ng + (g - g0) with g0 = g / gps * gps, ng = min(gps, Kmax - g0)
__global__ void __launch_bounds__(256) k_jk_linear(const double *__restrict__ P, unsigned Kmax, size_t N, const double *__restrict__ Mv,
                                                   float *__restrict__ out, unsigned W, unsigned gps)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const unsigned c = blockIdx.y; // replica
	out += (size_t)c * N;
	const double invK = 1. / Mv[c];
	double acc = 0;
	for (unsigned g = 0; g < Kmax; g++) {
		size_t row;
		if (gps) { const unsigned g0 = g / gps * gps, ng = (Kmax - g0) < gps ? (Kmax - g0) : gps; row = (size_t)g0 * W + (size_t)c * ng + (g - g0); }
		else row = (size_t)c * Kmax + g;
		const double v = P[row * N + n];
		acc = g ? acc + v : v;
	}
	out[n] = (float)(acc * invK);
}

static constexpr unsigned SIG_DELETED = ~0u; // (32-bit signatures: Kmax is an unsigned in t_tsPWS, any value is a legal group count)

// (host tables handed to asynchronous copies: the call synchronises its stream before they can change)
static int cs_done(hipStream_t st) { HIP_TRY(hipStreamSynchronize(st)); return 0; }

// Rows of the masked replicas: d_P[c * KM + g][N] (SCR_JKP; the replicas' trace counts follow the rows).  Always with room for one
// more column -- the plain groups, which the one-pass walk leaves behind the replicas -- so that the block does not move between
// the calls of one sharded jackknife.
static int replica_rows_buffer(tspws_hip_plan *pl, unsigned KM, unsigned C, double **d_P)
{
	void *v;
	int rc = scratch(pl, SCR_JKP, ((size_t)(C + 1) * KM * pl->N + C + 1) * sizeof(double), &v);
	if (rc) return rc;
	*d_P = (double *)v;
	return 0;
}

// Replicas [c_begin, c_end) from their partial-stack rows: transforms, phase stacks, weights, time-domain linear stacks,
// inverses.  Outputs land in rows c_begin.. of d_ls_out / d_ts_out / h_mtr_out ([C][N] arrays indexed by replica).
// Synchronises the stream before returning (host tables go out of scope).
static int finish_replicas(tspws_hip_plan *pl, const t_tsPWS *p, double *d_P, const std::vector<size_t> &Kc, unsigned C, unsigned c_begin,
                           unsigned c_end, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	const size_t N = pl->N;
	int rc;
	void *v;
	double *d_Mv = d_P + (size_t)C * KM * N;
	std::vector<double> h_Mv(C);
	for (unsigned c = 0; c < C; c++) h_Mv[c] = (double)Kc[c];
	HIP_TRY(hipMemcpyAsync(d_Mv, h_Mv.data(), C * sizeof(double), hipMemcpyHostToDevice, st));
	// Replicas are processed in batches: ONE forward launch transforms the KM partials of a whole batch of replicas (the
	// kernels fill the GPU far better with 100 traces than with 10), then per replica the phase accumulation and the
	// weight (KM, K_c), and the inverses two replicas at a time.
	const size_t nc = pl->ncoef;
	const unsigned nrep = c_end - c_begin;
	unsigned RB = (unsigned)std::max<size_t>(1, std::min<size_t>(nrep, tspws_part_budget_bytes() / std::max<size_t>(1, (size_t)KM * pl->npart * sizeof(double2))));
	if (RB > 1) RB &= ~1u; // pairs for the two-set inverse
	if ((rc = scratch(pl, SCR_PART, (size_t)RB * KM * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	// per replica of the batch: OUT (2 nc doubles) | ST | PS, then the reconstructions
	if ((rc = scratch(pl, SCR_JKOUT, ((size_t)RB * 6 * nc + (size_t)RB * N) * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *STr = OUT + (size_t)RB * 2 * nc, *xr = STr + (size_t)RB * 4 * nc;
	const bool fuse = tspws_fused_forward(pl);
	for (unsigned c0 = c_begin; c0 < c_end; c0 += RB) {
		const unsigned nr = std::min(RB, c_end - c0);
		// one slice of the fused forward kernel = the KM partial stacks of one replica: its stacks land in the replica's planes
		FuseOut fz;
		fz.accST = (double2 *)STr; fz.accPS = (double2 *)STr + nc; fz.stride = 2 * nc; fz.tps = KM;
		if ((rc = tspws_forward_parts_f64(pl, d_P + (size_t)c0 * KM * N, (size_t)nr * KM, N, part, st, fuse ? &fz : nullptr, ScaleRange()))) return rc;
		for (unsigned j = 0; j < nr; j++) h_mtr_out[c0 + j] = (unsigned)Kc[c0 + j];
		// the replicas of the batch side by side in every launch (grid.y): stacks of the scales the fused kernel left out,
		// weights with each replica's trace count, time-domain linear stacks, inverses two replicas per tap read, outputs
		FuseOut fj = fz; // replica j's slice went straight into its ST / PS planes
		tspws_launch_accumulate(pl, (const double2 *)part, KM, (double2 *)STr, (double2 *)STr + nc, 1, &fj, 1, st, nr, (size_t)KM * pl->npart, 2 * nc, nullptr, nullptr,
		                        ScaleRange());
		tspws_weight_batched(pl, (double2 *)OUT, (const double2 *)STr, (const double2 *)STr + nc, tspws_weight_mode(p->wu, p->unbiased, KM), (double)KM, p->wu,
		                     (const double *)(d_Mv + c0), nr, nc, 2 * nc, st);
		hipLaunchKernelGGL(k_jk_linear, dim3((unsigned)((N + 255) / 256), nr), dim3(256), 0, st, d_P + (size_t)c0 * KM * N, KM, N, (const double *)(d_Mv + c0),
		                   d_ls_out + (size_t)c0 * N, 0u, 0u);
		if ((rc = tspws_hip_inverse(pl, OUT, nr, xr, s))) return rc;
		tspws_epilogue_rows(d_ts_out + (size_t)c0 * N, xr, N, nr, st);
	}
	HIP_TRY(hipGetLastError());
	return cs_done(st); // host tables above go out of scope
}

// ------------------------------------------------------------------------------------------
// The masked replicas in stages (round 4).  Replica groups are contiguous in selected-trace order (ts_pws1f_lib.c:758-772,
// g = floor(k Kmax / K)), so once the traces up to the end of group g of EVERY column (replica / plain stack) have been
// streamed, row g of all columns is final: its forward transforms (the FP64-bound half of the call: 110 transforms at cfg4)
// run on a second stream while the next groups are streamed (the HBM-bound half).
//   stage s = groups [s gps, (s + 1) gps) of every column (ONE stage by default: what the two sides cost each other eats what the
//             overlap gains, see below), its traces = [T(s - 1), T(s)) with T(s) = the
//             end of the last trace that belongs to a group of stage <= s in any column;
//   streaming side: RUNNING sums.  The traces are cut into runs of one signature (= the group of the trace in every column);
//             a segment of a stage walks its runs in trace order without resetting its accumulators and stores a snapshot after
//             every run (k_prefix_walk, stream.hip).  A row -- group g of column c -- is the sum over the maximal stretches
//             [a, b) of consecutive runs that belong to it: sum of G(b) - G(a), G(k) = snapshot before run k (+ the final
//             snapshots of the earlier segments of its stage): ~10 signed snapshots per row (k_combine_terms) instead of a
//             chunk reduction + ~15 class rows per row -- 1.30 -> 1.17 ms for the streaming side of cfg4 (the 326 MB of snapshot
//             writes cost 0.25 ms of the 1.02-ms walk: with the stores aimed at two rows only it takes 0.78 ms).  With <= 16
//             columns the walk keeps a running sum per column in registers and stores the rows themselves (k_rows_walk +
//             k_seg_fix, stream.hip): 0.93 ms, no snapshots, no combining pass;
//   rows   are STAGE-major and, inside a stage, column-major: row(g, c) = g0 W + c ng + (g - g0)  (W = C [+ 1 for the plain
//             stack], g0 / ng = first group / groups of the stage), so the rows of a stage are one run and the ng rows of a
//             column in it are consecutive: the stage's forward launch is the FUSED kernel with one slice per column -- the
//             linear / phase stacks of the stage's groups stay in registers and leave one plane pair per (stage, column); the
//             scales with phase splits leave per-trace partials as always.  ONE accumulation at the end adds a column's
//             plane pairs in stage order and its split partials in group order -- the reference's order (:897-904) -- and
//             writes the weighted coefficients; the inverses of all columns share one batched launch.
// What the two halves cost each other (tools/probes/corun_probe.hip, tools/experiments/corun_fwd.py, profiles/r04_*cfg4*): a
// register-only FP64 stream keeps its rate beside a streaming kernel (which drops to 70 %), but k_fwd_lds beside the streaming
// side loses as much as the overlap gains: ten stages of one group each 2.82 ms (and 4.0 ms with fused slices of one trace),
// three stages 2.82, two 2.68-2.74, one (no overlap at all) 2.73-2.77; the round-3 call was 3.13 ms.  Stream / wave priorities,
// the cache policy of the streaming loads and the number of streaming workgroups change nothing (docs/history/round-4.md).
// Everything that depends on the selection only (runs, segments, term lists, row map, trace counts) is built once per
// selection (per host thread, keyed by content) AND stays on the device while the plan's table block is not reused
// (plan->jk_gen): a repeated selection issues no host-to-device copy at all.
// ------------------------------------------------------------------------------------------
struct MaskedPlan {
	// key
	size_t mtr = 0, N = 0, first = 0, mtr_local = 0;
	unsigned C = 0, KM = 0, gps = 0;
	bool with_main = false, valid = false, allow_direct = true;
	std::vector<char> sel;
	// products
	unsigned long long gen = 0;
	unsigned W = 0, nstage = 0;
	std::vector<size_t> Kc;
	std::vector<Chunk> runs;            // maximal runs of consecutive traces with one signature (cut at the stage ends), trace order
	std::vector<unsigned> seg_first;    // per stage: the first runs of its segments + the end (nseg + 1 entries), stages concatenated
	std::vector<unsigned> stage_seg0;   // per stage: its first entry in seg_first (nstage + 1)
	std::vector<unsigned> carry;        // per stage: snapshots whose sum is the prefix sum at the stage's start
	std::vector<unsigned> carry_ptr;    // (nstage + 1)
	std::vector<unsigned> trow_ptr;     // rows as signed sums of snapshots: [KM W + 1] pointers, rows in stage / column / group order
	std::vector<unsigned> tidx;
	std::vector<float> tcoef;
	// few columns (W <= tspws_rows_walk_wmax()): the rows straight from the walk -- per run the columns it belongs to and the columns
	// whose group ends with it (+ the rows those sums become)
	bool direct = false, unwritten = false; // unwritten: some row is never stored (an empty group): the row block is cleared first
	std::vector<RunDesc> rdesc;         // the runs with their column bits
	std::vector<unsigned> stage_run0;   // per stage: its first run (nstage + 1)
	std::vector<unsigned> stage_mid;    // per stage: first run of its second segment (== the next stage's first run: one segment)
	std::vector<unsigned> fix_row;      // [nstage][W]: the first row column c stores in the stage's second segment (~0u: none)
	std::vector<unsigned> flush_rows;   // flush destinations (RunDesc::frow points here)
	std::vector<unsigned> rowmap;       // [W][KM]: row of (column, group)
	std::vector<double> Mv;             // trace count per column (replicas: selected traces; plain stack: mtr)
	std::vector<char> blob;             // all device tables in one block, offsets below
	size_t o_mv = 0, o_rd = 0, o_tp = 0, o_ti = 0, o_tc = 0, o_map = 0, o_seg = 0, o_car = 0, o_fr = 0, o_fx = 0;
	unsigned row_of(unsigned g, unsigned c) const
	{
		const unsigned g0 = g / gps * gps, ng = std::min(gps, KM - g0);
		return g0 * W + c * ng + (g - g0);
	}
};

static unsigned long long next_masked_gen()
{
	static std::atomic<unsigned long long> g{0};
	return ++g;
}

// Trace shards: the signatures come from the WHOLE selection (a trace's group in a replica is its rank among all selected traces, :766),
// the runs cover the shard's traces [first, first + mtr_local) only, with trace indices local to the shard.
static const MaskedPlan &masked_plan(size_t N, size_t mtr, const char *h_sel, unsigned C, unsigned KM, bool with_main, unsigned gps, bool allow_direct,
                                     size_t first, size_t mtr_local)
{
	static thread_local MaskedPlan mp;
	if (mp.valid && mp.mtr == mtr && mp.N == N && mp.C == C && mp.KM == KM && mp.gps == gps && mp.with_main == with_main && mp.allow_direct == allow_direct && mp.first == first && mp.mtr_local == mtr_local &&
	    mp.sel.size() == (size_t)C * mtr && !memcmp(mp.sel.data(), h_sel, (size_t)C * mtr)) return mp;
	mp.valid = false;
	struct HostTimer { // TSPWS_JK_HOSTTIME=1: what a new selection costs the host (printed per rebuild)
		std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
		~HostTimer() { static const bool on = sweep_env("TSPWS_JK_HOSTTIME") != nullptr; if (on) printf("masked_plan: %.1f us of host work\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
	} host_timer;
	const unsigned W = C + (with_main ? 1u : 0u);
	mp.KM = KM; mp.gps = gps; mp.W = W;
	const unsigned nstage = (KM + gps - 1) / gps;
	// Signature of trace i: its group in every column (SIG_DELETED: not in that replica), COLUMN-major: one sequential pass per
	// column that also marks where a run ends (chg) and where each group of the column ends (for the stage ends).  The reference's
	// floor((double)(k * KM) / (double)Kc) (:766) is the integer quotient (k KM < 2^53 and a non-integer quotient is at least 1 / Kc away from the next integer), kept incrementally -- a 64-bit division
	// per trace and column was a third of the 0.3 ms a new selection cost the host.
	// Round 5: the columns as PIECES, not trace by trace.  A column's signature is piecewise constant: a replica's changes where its
	// selection byte changes and, inside a stretch of selected traces, where floor(k KM / Kc) steps (k = rank among the selected traces);
	// the plain stack's at ceil(g mtr / KM).  The stretches are found a machine word at a time, the group steps by ONE division per step
	// (<= KM per column), and only the pieces' starts are written anywhere: a new selection of cfg4 (10 replicas x 10 000 traces,
	// ~330 runs) costs the host ~0.03 ms instead of 0.16 (one pass per trace and column with a signature array of W x mtr words).
	mp.Kc.assign(C, 0);
	struct Piece { size_t pos; unsigned v; }; // the column has signature v from trace pos on (global index), up to the next piece
	static thread_local std::vector<std::vector<Piece>> pieces;
	static thread_local std::vector<unsigned char> chg; // chg[i]: trace i starts a run
	pieces.resize(W);
	chg.assign(mtr + 1, 0);
	const size_t lo = first, hi = first + mtr_local; // the shard
	if (lo < hi) chg[lo] = 1;
	std::vector<size_t> T(nstage, 0); // end of stage s: one past the last trace that belongs to a group of stage <= s in any column
	auto close_piece = [&](const Piece &pc, size_t end) { // [pc.pos, end) with one signature: stage ends inside the shard
		if (pc.v == SIG_DELETED || end <= lo || pc.pos >= hi) return;
		size_t &te = T[std::min(pc.v, KM - 1) / gps];
		const size_t e = std::min(end, hi) - lo;
		if (e > te) te = e;
	};
	for (unsigned c = 0; c < W; c++) {
		std::vector<Piece> &pc = pieces[c];
		pc.clear();
		auto emit = [&](size_t pos, unsigned v) {
			if (!pc.empty()) {
				if (pc.back().v == v) return;               // (no change after all)
				close_piece(pc.back(), pos);
			}
			pc.push_back(Piece{pos, v});
			if (pos) chg[pos] = 1;
		};
		if (c < C) {
			const unsigned char *row = (const unsigned char *)h_sel + (size_t)c * mtr;
			size_t n = 0;
			for (size_t i = 0; i < mtr; i++) n += row[i] == 1; // (vectorised)
			mp.Kc[c] = n;
			const unsigned long long Kc = std::max<size_t>(n, 1);
			unsigned long long k = 0, g = 0, kb = (Kc + KM - 1) / KM; // rank among the selected traces; its group; the rank at which the group steps next
			size_t i = 0;
			if (mtr && row[0] != 1) emit(0, SIG_DELETED);
			while (i < mtr) {
				if (row[i] != 1) { // a stretch that is not selected: up to the next byte 1
					const void *q = memchr(row + i, 1, mtr - i);
					i = q ? (size_t)((const unsigned char *)q - row) : mtr;
					continue;
				}
				size_t j = i; // a stretch of selected traces [i, j): words of eight bytes 1, then the tail
				while (j + 8 <= mtr) { unsigned long long w; memcpy(&w, row + j, 8); if (w != 0x0101010101010101ull) break; j += 8; }
				while (j < mtr && row[j] == 1) j++;
				const unsigned long long k1 = k + (j - i);
				if (kb <= k) { g = k * KM / Kc; kb = ((g + 1) * Kc + KM - 1) / KM; }
				emit(i, (unsigned)g);
				while (kb < k1) { // the group steps inside the stretch
					const size_t pos = i + (size_t)(kb - k);
					g = kb * KM / Kc; kb = ((g + 1) * Kc + KM - 1) / KM;
					emit(pos, (unsigned)g);
				}
				k = k1;
				if (j < mtr) emit(j, SIG_DELETED);
				i = j;
			}
		} else { // the plain stack: min(floor(i KM / mtr), KM - 1), steps at ceil(g mtr / KM)
			emit(0, 0);
			for (unsigned long long g = 1; g < KM; g++) {
				const unsigned long long pos = (g * mtr + KM - 1) / KM;
				if (pos >= mtr) break;
				emit((size_t)pos, (unsigned)(pos * KM / mtr));
			}
		}
		if (!pc.empty()) close_piece(pc.back(), mtr);
	}
	for (unsigned sg = 1; sg < nstage; sg++) T[sg] = std::max(T[sg], T[sg - 1]);
	T[nstage - 1] = mtr_local; // (traces past the last group of every column change nothing; they ride along)
	// runs of the shard (local trace indices), cut at signature changes and stage ends
	mp.runs.clear();
	std::vector<unsigned> run_stage;
	{
		unsigned sg = 0;
		for (size_t i = 0; i < mtr_local;) {
			while (sg + 1 < nstage && i >= T[sg]) sg++;
			const size_t lim = std::min(mtr_local, T[sg]);
			size_t j = i + 1;
			if (j < lim) { // the next trace of the shard that starts a run (bytes of chg: memchr)
				const void *q = memchr(chg.data() + lo + j, 1, lim - j);
				j = q ? (size_t)((const unsigned char *)q - chg.data()) - lo : lim;
			}
			Chunk c; c.t0 = i; c.count = (unsigned)(j - i); c.row = 0;
			if (c.count != j - i) { j = i + 0xFFFFFFF0ull; c.count = 0xFFFFFFF0u; } // (a run longer than 2^32 traces is cut)
			mp.runs.push_back(c);
			run_stage.push_back(sg);
			i = j;
		}
	}
	// group of run r in column c: the piece that holds the run's first trace (the columns' pieces and the runs are both in trace order)
	static thread_local std::vector<unsigned> rsig; // [W][runs]
	{
		const size_t nrr = mp.runs.size();
		rsig.resize((size_t)W * nrr);
		for (unsigned c = 0; c < W; c++) {
			const std::vector<Piece> &pc = pieces[c];
			size_t q = 0;
			for (size_t r = 0; r < nrr; r++) {
				const size_t t = lo + mp.runs[r].t0;
				while (q + 1 < pc.size() && pc[q + 1].pos <= t) q++;
				rsig[(size_t)c * nrr + r] = pc.empty() ? SIG_DELETED : pc[q].v;
			}
		}
	}
	auto SIG = [&](unsigned r, unsigned c) -> unsigned { return rsig[(size_t)c * mp.runs.size() + r]; }; // group of run r in column c
	const unsigned nr = (unsigned)mp.runs.size();
	mp.stage_run0.assign(nstage + 1, nr);
	for (unsigned r = nr; r-- > 0;) mp.stage_run0[run_stage[r]] = r;
	for (unsigned sg = nstage; sg-- > 0;) if (mp.stage_run0[sg] > mp.stage_run0[sg + 1]) mp.stage_run0[sg] = mp.stage_run0[sg + 1]; // (empty stages)
	mp.direct = allow_direct && W <= tspws_rows_walk_wmax() && W <= 32;
	mp.rdesc.clear(); mp.flush_rows.clear(); mp.unwritten = false;
	if (mp.direct) {
		std::vector<char> written((size_t)KM * W, 0);
		mp.rdesc.resize(nr);
		for (unsigned r = 0; r < nr; r++) { RunDesc d; memset(&d, 0, sizeof d); d.t0 = mp.runs[r].t0; d.count = mp.runs[r].count; mp.rdesc[r] = d; }
		// column by column, from the last run back: a run that belongs to the column ends the column's group when the next run that
		// belongs to it has another group (or there is none)
		for (unsigned c = 0; c < W; c++) {
			unsigned next_g = SIG_DELETED;
			for (unsigned r = nr; r-- > 0;) {
				const unsigned g = SIG(r, c);
				if (g == SIG_DELETED) continue;
				mp.rdesc[r].member |= 1u << c;
				if (g != next_g) { mp.rdesc[r].flush |= 1u << c; written[mp.row_of(std::min(g, KM - 1), c)] = 1; }
				next_g = g;
			}
		}
		for (unsigned r = 0; r < nr; r++) { // flush destinations in ascending column order
			mp.rdesc[r].frow = (unsigned)mp.flush_rows.size();
			for (unsigned c = 0; c < W; c++) if ((mp.rdesc[r].flush >> c) & 1u) mp.flush_rows.push_back(mp.row_of(std::min(SIG(r, c), KM - 1), c));
		}
		for (char w : written) if (!w) mp.unwritten = true;
		// two segments of similar trace counts per stage, and what the second one's first stores lack
		mp.stage_mid.assign(nstage, 0); mp.fix_row.assign((size_t)nstage * W, ~0u);
		for (unsigned sg = 0; sg < nstage; sg++) {
			const unsigned q0 = mp.stage_run0[sg], q1 = mp.stage_run0[sg + 1];
			size_t traces = 0, done = 0;
			for (unsigned r = q0; r < q1; r++) traces += mp.runs[r].count;
			unsigned qm = q1;
			for (unsigned r = q0; r < q1; r++) { if (r > q0 && 2 * done >= traces) { qm = r; break; } done += mp.runs[r].count; }
			mp.stage_mid[sg] = qm;
			for (unsigned r = qm; r < q1; r++) {
				unsigned fr = mp.rdesc[r].frow;
				for (unsigned c = 0; c < W; c++)
					if ((mp.rdesc[r].flush >> c) & 1u) { if (mp.fix_row[(size_t)sg * W + c] == ~0u) mp.fix_row[(size_t)sg * W + c] = mp.flush_rows[fr]; fr++; }
			}
		}
	}
	mp.seg_first.clear(); mp.stage_seg0.assign(nstage + 1, 0); mp.carry.clear(); mp.carry_ptr.assign(nstage + 1, 0);
	mp.trow_ptr.assign((size_t)KM * W + 1, 0); mp.tidx.clear(); mp.tcoef.clear();
	if (!mp.direct) { // the snapshot form: segments, carries, rows as signed sums of snapshots
	// segments: a stage's runs in ~256 / (column blocks) pieces of similar trace counts, each walked by its own workgroups
	static int seg_wgs = -1; // workgroups the streaming side aims at per stage (sweeps: TSPWS_JK_SEGWG)
	if (seg_wgs < 0) { const char *e = sweep_env("TSPWS_JK_SEGWG"); seg_wgs = e ? std::max(1, atoi(e)) : 256; }
	const unsigned bx = (unsigned)((N + 1023) / 1024), want_seg = std::max(1u, (unsigned)seg_wgs / std::max(1u, bx));
	mp.seg_first.clear(); mp.stage_seg0.assign(nstage + 1, 0);
	std::vector<unsigned> seg_of(nr, 0), seg_last;      // segment (global numbering) of a run; last run of a segment
	std::vector<unsigned> seg_stage_first(nstage + 1, 0); // first global segment of a stage
	{
		unsigned r = 0;
		for (unsigned sg = 0; sg < nstage; sg++) {
			mp.stage_seg0[sg] = (unsigned)mp.seg_first.size();
			seg_stage_first[sg] = (unsigned)seg_last.size();
			unsigned r1 = r;
			size_t traces = 0;
			while (r1 < nr && run_stage[r1] == sg) { traces += mp.runs[r1].count; r1++; }
			const unsigned nruns = r1 - r, nseg = std::min(want_seg, nruns);
			size_t done = 0;
			unsigned k = 0;
			for (unsigned q = r; q < r1; q++) {
				// run q opens segment k when the traces before it reach k / nseg of the stage
				if (k < nseg && (q == r || done * nseg >= (size_t)k * traces)) {
					if (q != r) seg_last.push_back(q - 1);
					mp.seg_first.push_back(q);
					k++;
				}
				seg_of[q] = (unsigned)(seg_stage_first[sg] + k - 1);
				done += mp.runs[q].count;
			}
			if (nruns) seg_last.push_back(r1 - 1);
			mp.seg_first.push_back(r1);
			r = r1;
		}
		mp.stage_seg0[nstage] = (unsigned)mp.seg_first.size();
		seg_stage_first[nstage] = (unsigned)seg_last.size();
	}
	// prefix sum at the start of a stage = sum of the final snapshots of the segments of the last non-empty stage before it
	mp.carry.clear(); mp.carry_ptr.assign(nstage + 1, 0);
	{
		std::vector<unsigned> cur;
		for (unsigned sg = 0; sg < nstage; sg++) {
			mp.carry_ptr[sg] = (unsigned)mp.carry.size();
			mp.carry.insert(mp.carry.end(), cur.begin(), cur.end());
			if (seg_stage_first[sg + 1] > seg_stage_first[sg]) {
				cur.clear();
				for (unsigned q = seg_stage_first[sg]; q < seg_stage_first[sg + 1]; q++) cur.push_back(seg_last[q]);
			}
		}
		mp.carry_ptr[nstage] = (unsigned)mp.carry.size();
	}
	// G(k) = sum of all traces before run k, as a signed sum of snapshots: snap[k - 1] + the final snapshots of the earlier segments
	// of the same stage (segment 0 of a stage starts from the carried prefix: its snapshots are global)
	auto add_G = [&](std::vector<std::pair<unsigned, int>> &terms, unsigned k, int sign) {
		if (!k) return;
		const unsigned r = k - 1, sg = run_stage[r];
		terms.emplace_back(r, sign);
		for (unsigned q = seg_stage_first[sg]; q < seg_of[r]; q++) terms.emplace_back(seg_last[q], sign);
	};
	// rows: for every column and group the maximal stretches [a, b) of consecutive runs that belong to it: sum of G(b) - G(a)
	const unsigned nrow = KM * W;
	std::vector<std::vector<std::pair<unsigned, int>>> lists(nrow);
	for (unsigned c = 0; c < W; c++) {
		unsigned a = 0, cur = SIG_DELETED;
		for (unsigned r = 0; r <= nr; r++) {
			const unsigned g = r < nr ? SIG(r, c) : SIG_DELETED;
			if (g == cur) continue;
			if (cur != SIG_DELETED) { auto &L = lists[mp.row_of(std::min(cur, KM - 1), c)]; add_G(L, r, +1); add_G(L, a, -1); }
			cur = g; a = r;
		}
	}
	mp.trow_ptr.assign((size_t)nrow + 1, 0); mp.tidx.clear(); mp.tcoef.clear();
	for (unsigned r = 0; r < nrow; r++) {
		mp.trow_ptr[r] = (unsigned)mp.tidx.size();
		auto &L = lists[r];
		std::sort(L.begin(), L.end());
		for (size_t i = 0; i < L.size();) { // merge equal snapshots, drop what cancels
			size_t j = i; int cf = 0;
			while (j < L.size() && L[j].first == L[i].first) cf += L[j++].second;
			if (cf) { mp.tidx.push_back(L[i].first); mp.tcoef.push_back((float)cf); }
			i = j;
		}
	}
	mp.trow_ptr[nrow] = (unsigned)mp.tidx.size();
	} // (snapshot form)
	mp.rowmap.assign((size_t)W * KM, 0);
	for (unsigned c = 0; c < W; c++) for (unsigned g = 0; g < KM; g++) mp.rowmap[(size_t)c * KM + g] = mp.row_of(g, c);
	mp.Mv.assign(W, 0.0);
	for (unsigned c = 0; c < C; c++) mp.Mv[c] = (double)mp.Kc[c];
	if (with_main) mp.Mv[C] = (double)(unsigned)mtr;
	// one block for the device: runs (16-byte records) | trace counts | run descriptors | term pointers | term snapshots | term
	// coefficients | row map | segments | carries | flush rows | fix rows -- ONE host-to-device copy per new selection
	{
		const size_t n_runs = mp.direct ? 0 : mp.runs.size(), n_tp = mp.trow_ptr.size(), n_t = mp.tidx.size(), n_map = mp.rowmap.size(), n_seg = mp.seg_first.size(),
		             n_car = mp.carry.size(), n_rd = mp.rdesc.size(), n_fr = mp.flush_rows.size(), n_fx = mp.fix_row.size();
		mp.o_mv = n_runs * sizeof(Chunk); mp.o_rd = mp.o_mv + W * sizeof(double); mp.o_tp = mp.o_rd + n_rd * sizeof(RunDesc); mp.o_ti = mp.o_tp + n_tp * 4;
		mp.o_tc = mp.o_ti + n_t * 4; mp.o_map = mp.o_tc + n_t * 4; mp.o_seg = mp.o_map + n_map * 4; mp.o_car = mp.o_seg + n_seg * 4; mp.o_fr = mp.o_car + n_car * 4;
		mp.o_fx = mp.o_fr + n_fr * 4;
		mp.blob.assign(mp.o_fx + std::max<size_t>(n_fx, 1) * 4, 0);
		char *b = mp.blob.data();
		if (n_runs) memcpy(b, mp.runs.data(), n_runs * sizeof(Chunk));
		memcpy(b + mp.o_mv, mp.Mv.data(), W * sizeof(double));
		if (n_rd) memcpy(b + mp.o_rd, mp.rdesc.data(), n_rd * sizeof(RunDesc));
		memcpy(b + mp.o_tp, mp.trow_ptr.data(), n_tp * 4);
		if (n_t) { memcpy(b + mp.o_ti, mp.tidx.data(), n_t * 4); memcpy(b + mp.o_tc, mp.tcoef.data(), n_t * 4); }
		memcpy(b + mp.o_map, mp.rowmap.data(), n_map * 4);
		if (n_seg) memcpy(b + mp.o_seg, mp.seg_first.data(), n_seg * 4);
		if (n_car) memcpy(b + mp.o_car, mp.carry.data(), n_car * 4);
		if (n_fr) memcpy(b + mp.o_fr, mp.flush_rows.data(), n_fr * 4);
		if (n_fx) memcpy(b + mp.o_fx, mp.fix_row.data(), n_fx * 4);
	}
	mp.mtr = mtr; mp.N = N; mp.C = C; mp.with_main = with_main; mp.nstage = nstage; mp.allow_direct = allow_direct; mp.first = first; mp.mtr_local = mtr_local;
	mp.sel.assign(h_sel, h_sel + (size_t)C * mtr);
	mp.gen = next_masked_gen();
	mp.valid = true;
	return mp;
}

// groups per stage: TSPWS_JK_GPS, else ceil(KM / stages) with TSPWS_JK_STAGES stages (default 1: no overlap -- see the table of stage counts above)
static unsigned masked_gps(unsigned KM)
{
	static int gps = -1, nst = -1;
	if (gps < 0) { const char *e = sweep_env("TSPWS_JK_GPS"); gps = e ? std::max(0, atoi(e)) : 0; }
	if (nst < 0) { const char *e = sweep_env("TSPWS_JK_STAGES"); nst = e ? std::max(1, atoi(e)) : 1; }
	if (gps > 0) return std::min((unsigned)gps, KM);
	return std::max(1u, (KM + (unsigned)nst - 1) / (unsigned)nst);
}

// 1: the pipelined call is possible for this plan / shape (per-trace coefficients of all KM W rows fit the scratch budget)
static bool masked_pipeline_ok(const tspws_hip_plan *pl, unsigned KM, unsigned W)
{
	static int off = -1;
	if (off < 0) { const char *e = sweep_env("TSPWS_JK_PIPELINE"); off = (e && *e == '0') ? 1 : 0; }
	if (off || tspws_generic_forward()) return false;
	return (size_t)KM * W * pl->npart * sizeof(double2) <= tspws_part_budget_bytes();
}

// device side of a MaskedPlan: the table block (uploaded when the plan's block does not hold this selection's already) and, for the
// direct walk, the [tail | end | carry] rows between its segments and stages
struct MaskedDev {
	char *tb = nullptr;
	double *carryblk = nullptr;
	bool have_carry = false;
	bool uploaded = false; // this call enqueued a copy of the memo's host tables: the call must not return before it has been read
};

static int masked_tables(tspws_hip_plan *pl, const MaskedPlan &mp, hipStream_t st, MaskedDev &dv)
{
	void *v;
	int rc;
	if ((rc = scratch(pl, SCR_JKTAB, mp.blob.size(), &v))) return rc;
	dv.tb = (char *)v;
	if (mp.direct) { if ((rc = scratch(pl, SCR_CLS, (size_t)3 * mp.W * mp.N * sizeof(double), &v))) return rc; dv.carryblk = (double *)v; }
	if (pl->jk_gen != mp.gen) {
		pl->jk_gen = 0;
		HIP_TRY(hipMemcpyAsync(dv.tb, mp.blob.data(), mp.blob.size(), hipMemcpyHostToDevice, st));
		pl->jk_gen = mp.gen;
		dv.uploaded = true;
	}
	dv.have_carry = false;
	return 0;
}

// the streaming side of stage sg: the stage's rows [g0 W, (g0 + ng) W) of d_rows from the shard's traces
static int masked_stream_stage(tspws_hip_plan *pl, const MaskedPlan &mp, MaskedDev &dv, const float *d_x, size_t ld, double *d_rows, unsigned sg, hipStream_t st)
{
	const size_t N = mp.N;
	const unsigned W = mp.W, KM = mp.KM, nrow = KM * W;
	const unsigned g0 = sg * mp.gps, ng = std::min(mp.gps, KM - g0), r0 = g0 * W, r1 = r0 + ng * W;
	int rc;
	if (mp.direct) {
		// few columns: the rows themselves from the walk (a running sum per column in registers; k_rows_walk)
		if (sg == 0 && mp.unwritten) HIP_TRY(hipMemsetAsync(d_rows, 0, (size_t)nrow * N * sizeof(double), st));
		const unsigned q0 = mp.stage_run0[sg], q1 = mp.stage_run0[sg + 1];
		if (q1 > q0) {
			// what the VEC4 walk takes for granted (stream.hip: ONE load stream per segment from the first run's t0 over the runs' total, the loaders'
			// and the writer wave's barrier counts from the same descriptors): inside a segment the runs tile consecutive traces, none is empty
			const unsigned qm_ = (mp.stage_mid[sg] > q0 && mp.stage_mid[sg] < q1) ? mp.stage_mid[sg] : q1;
			for (unsigned q = q0; q < q1; q++) {
				const RunDesc &d = mp.rdesc[q];
				if (!d.count || (q + 1 < q1 && q + 1 != qm_ && mp.rdesc[q + 1].t0 != d.t0 + d.count))
					return fail(TSPWS_E_ARG, "masked stack: the runs of a walk segment do not tile consecutive traces");
			}
			if ((rc = tspws_rows_walk_launch(d_x, ld, N, (const RunDesc *)(dv.tb + mp.o_rd), q0, mp.stage_mid[sg], q1, W, (const unsigned *)(dv.tb + mp.o_fr),
			                                 (const unsigned *)(dv.tb + mp.o_fx) + (size_t)sg * W, d_rows, dv.carryblk, dv.have_carry ? 1 : 0, sg + 1 < mp.nstage ? 1 : 0, st)))
				return rc;
			dv.have_carry = true;
		}
		return 0;
	}
	// running sums over the stage's traces (snapshots after every run), then its rows as signed sums of snapshots
	const unsigned k0 = mp.stage_seg0[sg], nseg = mp.stage_seg0[sg + 1] - k0 - 1;
	double *d_snap = nullptr; size_t ldpc = 0;
	if ((rc = tspws_prefix_launch(pl, d_x, ld, N, (const Chunk *)dv.tb, (const unsigned *)(dv.tb + mp.o_seg) + k0, nseg, mp.runs.size(),
	                              (const unsigned *)(dv.tb + mp.o_car) + mp.carry_ptr[sg], mp.carry_ptr[sg + 1] - mp.carry_ptr[sg], &d_snap, &ldpc, st)))
		return rc;
	tspws_combine_terms_launch(d_snap, ldpc, (const unsigned *)(dv.tb + mp.o_tp) + r0, (const unsigned *)(dv.tb + mp.o_ti), (const float *)(dv.tb + mp.o_tc), r1 - r0,
	                           d_rows + (size_t)r0 * N, N, st);
	return 0;
}

static bool masked_allow_direct()
{
	static int no_direct = -1; // TSPWS_JK_DIRECT=0: the snapshot form also for few columns (tests, A/B)
	if (no_direct < 0) { const char *e = sweep_env("TSPWS_JK_DIRECT"); no_direct = (e && *e == '0') ? 1 : 0; }
	return !no_direct;
}

// The rows of every column from ONE pass over a shard, replica-major: d_rows[(c KM + g) N] for the replicas, the plain groups of
// ALL traces (with_main, ts_pws1f_lib.c:876) as column C behind them -- one stage of KM groups, where row(g, c) = c KM + g.
// Sharded ensembles: d_x holds traces [first, first + mtr_local) of the mtr the selection refers to; the rows of all shards add up
// to the rows of the whole ensemble.  (The parts-budget fallback of the one-device call and the sharded jackknife.)
static int masked_rows(tspws_hip_plan *pl, unsigned KM, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C, bool with_main,
                       size_t first, size_t mtr_local, hipStream_t st, double **d_rows, const MaskedPlan **plan)
{
	if (first > mtr || mtr_local > mtr - first) return fail(TSPWS_E_ARG, "masked rows: shard outside the ensemble");
	const MaskedPlan &mp = masked_plan(pl->N, mtr, h_sel, C, KM, with_main, KM, masked_allow_direct(), first, mtr_local);
	int rc;
	if ((rc = replica_rows_buffer(pl, KM, C, d_rows))) return rc;
	MaskedDev dv;
	if ((rc = masked_tables(pl, mp, st, dv)) || (rc = masked_stream_stage(pl, mp, dv, d_x, ld, *d_rows, 0, st))) return rc;
	*plan = &mp;
	return 0;
}

static int masked_two_stage_pipelined(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C,
                                      float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s, bool with_stack, float *d_ls, float *d_ts)
{
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	const size_t N = pl->N, nc = pl->ncoef;
	const MaskedPlan &mp = masked_plan(N, mtr, h_sel, C, KM, with_stack, masked_gps(KM), masked_allow_direct(), 0, mtr);
	const unsigned W = mp.W, nrow = KM * W;
	int rc;
	void *v;
	// device blocks: partial-stack rows, split partials, slice planes, stacks + weighted sets + reconstructions (the snapshots of
	// the streaming side live in the chunk block)
	const bool fuse = tspws_fused_forward(pl);
	if ((rc = scratch(pl, SCR_JKP, ((size_t)nrow * N + W) * sizeof(double), &v))) return rc;
	double *d_rows = (double *)v;
	if ((rc = scratch(pl, SCR_PART, (size_t)nrow * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	double2 *planes = nullptr; // [stage][column][ST | PS]
	if (fuse) { if ((rc = scratch(pl, SCR_FZ, (size_t)mp.nstage * W * 2 * nc * sizeof(double2), &v))) return rc; planes = (double2 *)v; }
	const unsigned nrec = W + (with_stack ? 1u : 0u); // reconstructions: OUT of every column (+ ST of the plain stack)
	if ((rc = scratch(pl, SCR_JKOUT, ((size_t)(nrec + 2 * W) * 2 * nc + (size_t)nrec * N) * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *STr = OUT + (size_t)nrec * 2 * nc, *xr = STr + (size_t)W * 4 * nc;
	MaskedDev dv;
	if ((rc = masked_tables(pl, mp, st, dv))) return rc;
	double *d_Mv = (double *)(dv.tb + mp.o_mv);
	const unsigned *d_map = (const unsigned *)(dv.tb + mp.o_map);
	if (!pl->xf) {
		int lo = 0, hi = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi)); // (lo = least urgent)
		const char *e = sweep_env("TSPWS_JK_XFPRIO");
		HIP_TRY(hipStreamCreateWithPriority(&pl->xf, hipStreamNonBlocking, e ? (atoi(e) > 0 ? hi : lo) : 0));
	}
	while (pl->stage_ev.size() < mp.nstage + 1) {
		hipEvent_t e;
		HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
		pl->stage_ev.push_back(e);
	}
	const unsigned nbx = (unsigned)((N + 255) / 256);
	// the last stage's forward kernel completes the stacks itself (TSPWS_JK_FINAL: 0 never, 1 always; default: with ONE stage only -- with
	// earlier stages' plane pairs to add in front the time just moves from the accumulation into the kernel's epilogue)
	static int fin_env = -2;
	if (fin_env == -2) { const char *e = sweep_env("TSPWS_JK_FINAL"); fin_env = e ? atoi(e) : -1; }
	const bool fin_in_kernel = fin_env < 0 ? mp.nstage == 1 : fin_env != 0;
	unsigned last_spec_first = pl->S; // scales [.., S) of the last stage were completed by the spectral engine (forward.hip)
	bool early_inv = false;           // the inverses of the FIR kernels' octaves were launched behind those kernels (tspws_inverse_pairs_early)
	unsigned early_split = 0;
	bool lin_mid = false;             // ... and its chain recorded ev_mid behind itself
	// stream of the transforms: the second stream, so that a stage's transforms run beside the next stage's walk -- with ONE stage (the default) the
	// caller's own: nothing runs beside the walk then, and every hand-over to another stream costs ~20 us before the first kernel there starts
	const hipStream_t fs = mp.nstage == 1 ? st : pl->xf;
	for (unsigned sg = 0; sg < mp.nstage; sg++) {
		// HBM-bound half of the stage on the caller's stream: its rows from its traces
		if ((rc = masked_stream_stage(pl, mp, dv, d_x, ld, d_rows, sg, st))) return rc;
		const unsigned g0 = sg * mp.gps, ng = std::min(mp.gps, KM - g0), r0 = g0 * W, r1 = r0 + ng * W;
		HIP_TRY(hipEventRecord(pl->stage_ev[sg], st));
		// FP64-bound half on the transforms' stream: the stage's rows, one slice of ng rows per column
		if (fs != st) HIP_TRY(hipStreamWaitEvent(fs, pl->stage_ev[sg], 0));
		FuseOut fz;
		if (fuse) { fz.accST = planes + (size_t)sg * W * 2 * nc; fz.accPS = fz.accST + nc; fz.stride = 2 * nc; fz.tps = ng; }
		unsigned stage_spec_first = pl->S;
		if (fuse && fin_in_kernel && sg + 1 == mp.nstage) {
			// the last stage completes the columns' stacks in the forward kernel itself: earlier stages' plane pairs in front, weights on
			// the spot -- no pass over the fused scales (98 % of the coefficients) afterwards
			FuseFinal &f = fz.fin;
			f.pST = planes; f.pPS = planes + nc; f.pair_stride = (size_t)W * 2 * nc; f.slice_stride = 2 * nc; f.nprev = sg;
			f.OUT = (double2 *)OUT; f.out_stride = nc; f.Mv = d_Mv; f.mode = tspws_weight_mode(p->wu, p->unbiased, KM); f.K = (double)KM; f.wu = p->wu;
			f.keep_slice = with_stack ? (int)C : -1; f.keepST = (double2 *)OUT + (size_t)W * nc;
			fz.allow_spec = true; // (the far-decimated octaves of these ng W rows may go through the spectral engine: forward.hip decides)
		}
		if (fuse && C && sg + 1 == mp.nstage) { // (the linear stacks of the replicas may start behind the spectral chain's transposition, beside the transforms)
			if (!pl->ev_mid) HIP_TRY(hipEventCreateWithFlags(&pl->ev_mid, hipEventDisableTiming | hipEventDisableSystemFence));
			fz.ev_mid = pl->ev_mid;
		}
		// The inverses in two halves (one stage, in-kernel completion, pairs of reconstructions): the octaves of the scales the FIR kernels complete
		// start behind THEM on their stream, beside the tail of the spectral chain -- a run of short, latency-bound kernels that ends 0.2 ms after
		// the FIR kernels at cfg4 -- instead of behind the chain; only the chain's own octaves and the combining kernel are left for the end.
		// TSPWS_JK_EARLY_INV=0 (sweeps): all inverses behind the chain
		static const bool early_off = sweep_env("TSPWS_JK_EARLY_INV") && !strcmp(sweep_env("TSPWS_JK_EARLY_INV"), "0");
		const bool want_early = fuse && fz.fin.OUT && mp.nstage == 1 && !(nrec & 1u) && !early_off;
		fz.defer_fir_join = want_early;
		pl->le.ready = pl->stage_ev[sg]; // the producer's own event: the forward launch's other streams wait for it directly, not for a re-record on xf
		rc = tspws_forward_parts_f64(pl, d_rows + (size_t)r0 * N, r1 - r0, N, part + (size_t)r0 * pl->npart, fs, fuse ? &fz : nullptr, ScaleRange());
		pl->le.ready = nullptr;
		if (rc) return rc;
		if (fz.fir_stream) { // the FIR kernels are still un-joined on their stream
			// (every scale below the chain's must be complete behind the FIR kernels: fused and completed in the kernel -- no pass of k_accumulate_parts)
			if (fz.spec_first < pl->S && tspws_first_unfused_scale(pl) >= fz.spec_first)
				if ((rc = tspws_inverse_pairs_early(pl, (const double2 *)OUT, nrec / 2, fz.spec_first, fz.fir_stream, &early_inv))) return rc;
			if ((rc = tspws_join_fir_stream(pl, fz.fir_stream, fs))) return rc;
			early_split = fz.spec_first;
		}
		if (fuse) stage_spec_first = fz.spec_first;
		last_spec_first = stage_spec_first;
		lin_mid = fuse && fz.mid_recorded;
	}
	if (fs != st) HIP_TRY(hipEventRecord(pl->stage_ev[mp.nstage], fs));
	// time-domain linear stacks of the replicas (:799-811) while the last transforms run -- unless those start with the spectral chain's
	// transposition of the same rows: two kernels that walk the rows at a 1-MB stride at the same time take 0.57 + 0.93 ms instead of
	// 0.03 + 0.06 (cfg4), so the linear stacks then wait for that transposition (ev_mid)
	const bool lin_first = last_spec_first >= pl->S;
	bool lin_side = false; // the linear stacks went to the second stream: joined below
	if (C && lin_first && fs != st) hipLaunchKernelGGL(k_jk_linear, dim3(nbx, C), dim3(256), 0, st, (const double *)d_rows, KM, N, (const double *)d_Mv, d_ls_out, W, mp.gps);
	else if (C && (lin_mid || lin_first)) { // beside the transforms on the second stream: behind the chain's transposition, or (no chain) behind the walk
		HIP_TRY(hipStreamWaitEvent(pl->xf, lin_mid ? pl->ev_mid : pl->stage_ev[mp.nstage - 1], 0));
		hipLaunchKernelGGL(k_jk_linear, dim3(nbx, C), dim3(256), 0, pl->xf, (const double *)d_rows, KM, N, (const double *)d_Mv, d_ls_out, W, mp.gps);
		if (!pl->ev_lin) HIP_TRY(hipEventCreateWithFlags(&pl->ev_lin, hipEventDisableTiming | hipEventDisableSystemFence));
		HIP_TRY(hipEventRecord(pl->ev_lin, pl->xf));
		lin_side = true;
	}
	if (fs != st) HIP_TRY(hipStreamWaitEvent(st, pl->stage_ev[mp.nstage], 0));
	if (C && !lin_first && !lin_mid) hipLaunchKernelGGL(k_jk_linear, dim3(nbx, C), dim3(256), 0, st, (const double *)d_rows, KM, N, (const double *)d_Mv, d_ls_out, W, mp.gps);
	// stacks of every column: its plane pairs in stage order + its split partials in group order; weights by the same launch
	// (K = KM, M = the column's traces)
	WeightArgs wa;
	wa.OUT = (double2 *)OUT; wa.out_stride = nc; wa.mode = tspws_weight_mode(p->wu, p->unbiased, KM); wa.K = (double)KM; wa.wu = p->wu; wa.Mv = d_Mv;
	wa.planes_batch = with_stack ? (int)C : -2; // only the plain stack's ST is reconstructed (ls); the replicas need their weighted coefficients only
	AccExtra ex;
	ex.rowmap = d_map; ex.y_fz = 2 * nc;
	FuseOut fa;
	if (fuse) { fa.accST = planes; fa.accPS = planes + nc; fa.stride = (size_t)W * 2 * nc; fa.applied = true; }
	if (fuse && fin_in_kernel) {
		// only the scales with phase splits are left; the plain stack's ST goes straight to its set behind the weighted ones (batch C: ST + C y_stack)
		ex.fused_done = true;
		double2 *ST_arg = with_stack ? (double2 *)OUT + (size_t)W * nc - (size_t)C * 2 * nc : (double2 *)STr;
		// (the launch starts at the first scale the fused kernel left out -- the split scales are the far-decimated ones at the end of the
		// list --; fused scales inside its range are skipped by the kernel)
		const unsigned s_first = tspws_first_unfused_scale(pl), s_end = std::min(pl->S, last_spec_first); // (the spectral scales are complete)
		ScaleRange rg; rg.s0 = s_first; rg.s1 = s_end;
		if (s_first == 0 && s_end == pl->S) rg = ScaleRange(); // (no fused scale in front, none taken behind: everything)
		if (s_first < s_end) tspws_launch_accumulate(pl, (const double2 *)part, KM, ST_arg, (double2 *)STr + nc, 1, &fa, mp.nstage, st, W, 0, 2 * nc, nullptr, &wa, rg, &ex);
	} else {
		tspws_launch_accumulate(pl, (const double2 *)part, KM, (double2 *)STr, (double2 *)STr + nc, 1, fuse ? &fa : nullptr, mp.nstage, st, W, 0, 2 * nc, nullptr, &wa, ScaleRange(), &ex);
		if (with_stack) HIP_TRY(hipMemcpyAsync(OUT + (size_t)W * 2 * nc, STr + (size_t)C * 4 * nc, nc * sizeof(double2), hipMemcpyDeviceToDevice, st)); // ST of the plain stack: the last set
	}
	if (early_inv) { if ((rc = tspws_inverse_pairs_late(pl, (const double2 *)OUT, xr, nrec / 2, early_split, st))) return rc; }
	else if ((rc = tspws_hip_inverse(pl, OUT, nrec, xr, s))) return rc;
	if (C) tspws_epilogue_rows(d_ts_out, xr, N, C, st);
	if (with_stack && (rc = tspws_hip_epilogue(d_ls, d_ts, xr + (size_t)(C + 1) * N, xr + (size_t)C * N, N, (unsigned)mtr, s))) return rc;
	for (unsigned c = 0; c < C; c++) h_mtr_out[c] = (unsigned)mp.Kc[c];
	if (lin_side) HIP_TRY(hipStreamWaitEvent(st, pl->ev_lin, 0)); // (long done)
	HIP_TRY(hipGetLastError());
	// A call that uploaded the memo's tables waits for its stream (the host block may be rebuilt by the next call).  A call that found them
	// on the device -- the same selection as last time: the steady state of a caller that stacks many ensembles with the same time stamps --
	// returns like tspws_hip_stack does, with its kernels in flight: h_mtr_out is host data, the device outputs are stream-ordered.  (The
	// synchronisation cost such a loop ~0.13 ms of idle GPU per call at cfg4: wake-up, return, re-entry and the first launches of the next call.)
	return dv.uploaded ? cs_done(st) : 0;
}

// All C masked two-stage replicas from ONE pass over the traces (shared by the jackknife and the two-stage
// random subsampling, whose per-replica bodies are identical in the reference: :758-811 and :642-691).
// with_stack: the two-stage stack of ALL traces as well, from the same pass (the reference walks the traces 1 + C times).
static int masked_two_stage(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C,
                            float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s, bool with_stack, float *d_ls, float *d_ts)
{
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	int rc;
	if (masked_pipeline_ok(pl, KM, C + (with_stack ? 1u : 0u))) {
		rc = masked_two_stage_pipelined(pl, p, d_x, ld, mtr, h_sel, C, d_ls_out, d_ts_out, h_mtr_out, s, with_stack, d_ls, d_ts);
		if (rc) { (void)hipStreamSynchronize(pl->xf); (void)cs_done(st); }
		return rc;
	}
	// over the parts budget (or TSPWS_JK_PIPELINE=0): the same one-pass rows, then the replicas in batches that fit
	double *d_P;
	const MaskedPlan *mp;
	if ((rc = masked_rows(pl, KM, d_x, ld, mtr, h_sel, C, with_stack, 0, mtr, st, &d_P, &mp))) { (void)cs_done(st); return rc; }
	if (with_stack) {
		double *main_rows; size_t nd;
		if ((rc = tspws_hip_reduce_buffer(pl, p, mtr, &main_rows, &nd))) { (void)cs_done(st); return rc; }
		HIP_TRY(hipMemcpyAsync(main_rows, d_P + (size_t)C * KM * pl->N, (size_t)KM * pl->N * sizeof(double), hipMemcpyDeviceToDevice, st));
		if ((rc = tspws_hip_stack_finish(pl, p, mtr, d_ls, d_ts, s))) { (void)cs_done(st); return rc; }
	}
	const std::vector<size_t> Kc = mp->Kc;
	rc = finish_replicas(pl, p, d_P, Kc, C, 0, C, d_ls_out, d_ts_out, h_mtr_out, s);
	(void)cs_done(st);
	return rc;
}

extern "C" int tspws_hip_jackknife(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel,
                                   unsigned C, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	if (!pl || !p || !d_x || !h_sel || !d_ls_out || !d_ts_out || !h_mtr_out) return fail(TSPWS_E_ARG, "jackknife: NULL");
	if (!is_two_stage(p, mtr) || !C) return 0; // single-stage variant is an empty stub in the reference (:711-716)
	return masked_two_stage(pl, p, d_x, ld, mtr, h_sel, C, d_ls_out, d_ts_out, h_mtr_out, s, false, nullptr, nullptr);
}

// The two-stage stack AND its C jackknife replicas from ONE pass over the device-resident traces: tspws_hip_stack followed by
// tspws_hip_jackknife, with the traces streamed once instead of twice (the reference: 1 + C times, :216 and :758-772).  The
// stack's groups are then sums of class sums instead of chunk sums: the FP64 rounding of the partial stacks differs in the
// last bits from tspws_hip_stack's.
extern "C" int tspws_hip_stack_jackknife(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, float *d_ls, float *d_ts,
                                         const char *h_sel, unsigned C, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	if (!pl || !p || !d_x || !mtr || !d_ls || !d_ts) return fail(TSPWS_E_ARG, "stack_jackknife: bad argument");
	if (!is_two_stage(p, mtr) || !C || !h_sel) return tspws_hip_stack(pl, p, d_x, ld, mtr, d_ls, d_ts, s); // no replicas to share the pass with
	if (!d_ls_out || !d_ts_out || !h_mtr_out) return fail(TSPWS_E_ARG, "stack_jackknife: NULL replica outputs");
	return masked_two_stage(pl, p, d_x, ld, mtr, h_sel, C, d_ls_out, d_ts_out, h_mtr_out, s, true, d_ls, d_ts);
}

// ---- trace-sharded jackknife: shard-local rows -> the caller's reduction -> replicas finished where they are owned ----
extern "C" int tspws_hip_jackknife_buffer(tspws_hip_plan *pl, const t_tsPWS *p, unsigned C, double **d_buf, size_t *nd)
{
	if (!pl || !p || !d_buf || !nd) return fail(TSPWS_E_ARG, "jackknife_buffer: NULL");
	if (!p->Kmax || !C) return fail(TSPWS_E_ARG, "jackknife_buffer: two-stage calls with C > 0");
	HIP_TRY(hipSetDevice(pl->device));
	*nd = (size_t)C * p->Kmax * pl->N;
	return replica_rows_buffer(pl, p->Kmax, C, d_buf);
}

extern "C" int tspws_hip_jackknife_local(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                         size_t mtr_global, const char *h_sel, unsigned C, void *s)
{
	if (!pl || !p || !h_sel || (!d_x && mtr_local)) return fail(TSPWS_E_ARG, "jackknife_local: NULL");
	if (!is_two_stage(p, mtr_global) || !C) return fail(TSPWS_E_ARG, "jackknife_local: two-stage calls with C > 0");
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	double *main_rows, *d_P;
	size_t nd;
	int rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &main_rows, &nd))) return rc;
	if ((rc = replica_rows_buffer(pl, KM, C, &d_P))) return rc;
	// ONE pass over the shard for the plain groups and for every replica (empty shard: all rows zero)
	const MaskedPlan *mp;
	if ((rc = masked_rows(pl, KM, d_x, ld, mtr_global, h_sel, C, true, first, mtr_local, st, &d_P, &mp))) { (void)cs_done(st); return rc; }
	HIP_TRY(hipMemcpyAsync(main_rows, d_P + (size_t)C * KM * pl->N, (size_t)KM * pl->N * sizeof(double), hipMemcpyDeviceToDevice, st));
	return cs_done(st);
}

extern "C" int tspws_hip_jackknife_finish(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, const char *h_sel, unsigned C, unsigned c_begin,
                                          unsigned c_end, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	if (!pl || !p || !h_sel || !d_ls_out || !d_ts_out || !h_mtr_out) return fail(TSPWS_E_ARG, "jackknife_finish: NULL");
	if (!is_two_stage(p, mtr_global) || !C || c_begin > c_end || c_end > C) return fail(TSPWS_E_ARG, "jackknife_finish: two-stage calls, 0 <= c_begin <= c_end <= C");
	if (c_begin == c_end) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	std::vector<size_t> Kc(C, 0);
	for (unsigned c = 0; c < C; c++) for (size_t i = 0; i < mtr_global; i++) if (h_sel[(size_t)c * mtr_global + i] == 1) Kc[c]++;
	double *d_P;
	int rc;
	if ((rc = replica_rows_buffer(pl, p->Kmax, C, &d_P))) return rc;
	return finish_replicas(pl, p, d_P, Kc, C, c_begin, c_end, d_ls_out, d_ts_out, h_mtr_out, s);
}

// ------------------------------------------------------------------------------------------
// random subsampling (SubsamplingPlan :355-383, tspws_subsmpl_float :501-610, TwoStage_subsmpl_float :612-709)
// ------------------------------------------------------------------------------------------
extern "C" int tspws_subsampling_plan(char *sel, size_t J, size_t K)
{
	if (!sel) return 1;
	if (K > J) return 2;
	size_t k = 0;
	if (2 * K < J) { // fewer ones than zeros: switch ones on
		memset(sel, 0, J);
		while (k < K) { const size_t j = (size_t)rand() % J; if (!sel[j]) { k++; sel[j] = 1; } }
	} else {         // otherwise switch zeros on
		memset(sel, 1, J);
		K = J - K;
		while (k < K) { const size_t j = (size_t)rand() % J; if (sel[j]) { k++; sel[j] = 0; } }
	}
	return 0;
}

// ST_m += Y_b, PS_m += Y_b/|Y_b| for every mask m that contains trace b; one thread per coefficient.  The running stacks of up
// to 8 masks (blockIdx.y = group of 8 masks) stay in registers while the thread walks ALL traces of the batch in order: a trace is
// summed over its splits and normalised once and goes to the masks that select it -- the planes are read and written once per
// batch, not once per 8 traces.
__global__ void __launch_bounds__(256) k_accumulate_masked(const double2 *__restrict__ part, size_t npart, const ScaleDesc *__restrict__ sc,
                                                           unsigned S, size_t ncoef, unsigned ntr, const char *__restrict__ sel, size_t mtr,
                                                           size_t t0, unsigned M, double2 *__restrict__ ST, double2 *__restrict__ PS)
{
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].acc_off <= blockIdx.x) lo = mid; else hi = mid;
	}
	const unsigned Ns = sc[lo].Ns, nsplit = sc[lo].nsplit;
	const unsigned k = (blockIdx.x - sc[lo].acc_off) * 256 + threadIdx.x;
	if (k >= Ns) return;
	const size_t i = sc[lo].coef_off + k;
	const double2 *p0 = part + sc[lo].part_off + k;
	const unsigned m0 = blockIdx.y * 8u, nm = (M - m0) < 8u ? (M - m0) : 8u;
	double2 st[8], ps[8];
	bool any[8];
#pragma unroll
	for (int m = 0; m < 8; m++) {
		any[m] = false;
		if ((unsigned)m < nm) { st[m] = ST[(size_t)(m0 + m) * ncoef + i]; ps[m] = PS[(size_t)(m0 + m) * ncoef + i]; }
		else { st[m] = make_double2(0, 0); ps[m] = make_double2(0, 0); }
	}
	for (unsigned b0 = 0; b0 < ntr; b0 += 4) { // four traces at a time: their loads are independent, the additions stay in trace order
		double2 a[4], u[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			a[j] = make_double2(0, 0);
			if (b0 + (unsigned)j < ntr) a[j] = p0[(size_t)(b0 + (unsigned)j) * npart];
		}
		for (unsigned sp = 1; sp < nsplit; sp++) {
			double2 t[4];
#pragma unroll
			for (int j = 0; j < 4; j++) t[j] = (b0 + (unsigned)j < ntr) ? p0[(size_t)(b0 + (unsigned)j) * npart + (size_t)sp * Ns] : make_double2(0, 0);
#pragma unroll
			for (int j = 0; j < 4; j++) { a[j].x += t[j].x; a[j].y += t[j].y; }
		}
#pragma unroll
		for (int j = 0; j < 4; j++) { u[j] = make_double2(0, 0); add_unit_phasor(u[j], a[j]); }
#pragma unroll
		for (int j = 0; j < 4; j++) {
			if (b0 + (unsigned)j < ntr) {
#pragma unroll
				for (int m = 0; m < 8; m++) {
					if ((unsigned)m < nm && sel[(size_t)(m0 + m) * mtr + t0 + b0 + (unsigned)j] == 1) { // (wave-uniform)
						st[m].x += a[j].x; st[m].y += a[j].y; ps[m].x += u[j].x; ps[m].y += u[j].y; any[m] = true;
					}
				}
			}
		}
	}
#pragma unroll
	for (int m = 0; m < 8; m++)
		if ((unsigned)m < nm && any[m]) { ST[(size_t)(m0 + m) * ncoef + i] = st[m]; PS[(size_t)(m0 + m) * ncoef + i] = ps[m]; }
}

// time-domain linear stacks of the subsamples with the reference's FLOAT accumulator, traces in order
// (ts_pws1f_lib.c:538-542), then the float scale W/K (:579-583).  grid.y = mask
__global__ void __launch_bounds__(256) k_sub_linear(const float *__restrict__ x, size_t ld, size_t N, size_t mtr, const char *__restrict__ sel,
                                                    float scale, float *__restrict__ out)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const char *row = sel + (size_t)blockIdx.y * mtr;
	float acc = 0.f;
	for (size_t i0 = 0; i0 < mtr; i0 += 8) { // eight rows' loads in flight; the additions keep the trace order
		float v[8];
		bool on[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			on[j] = i0 + (size_t)j < mtr && row[i0 + (size_t)j] == 1; // (wave-uniform)
			v[j] = on[j] ? x[(i0 + (size_t)j) * ld + n] : 0.f;
		}
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (on[j]) acc = (float)((double)acc + (double)v[j]);
	}
	out[(size_t)blockIdx.y * N + n] = acc * scale;
}

extern "C" int tspws_hip_subsample(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, unsigned M,
                                   float *d_ls_out, float *d_ts_out, void *s)
{
	if (!pl || !p || !d_x || !d_ls_out || !d_ts_out) return fail(TSPWS_E_ARG, "subsample: NULL");
	if (!M || !mtr) return 0;
	const size_t K = (size_t)ceil((double)mtr * p->subsmpl_p);
	std::vector<char> sel((size_t)M * mtr);
	for (unsigned m = 0; m < M; m++) tspws_subsampling_plan(sel.data() + (size_t)m * mtr, mtr, K); // same rand() order as the reference
	return tspws_hip_subsample_sel(pl, p, d_x, ld, mtr, M, sel.data(), d_ls_out, d_ts_out, s);
}

extern "C" int tspws_hip_subsample_sel(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, unsigned M,
                                       const char *h_sel, float *d_ls_out, float *d_ts_out, void *s)
{
	if (!pl || !p || !d_x || !h_sel || !d_ls_out || !d_ts_out) return fail(TSPWS_E_ARG, "subsample: NULL");
	if (!M || !mtr) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const size_t K = (size_t)ceil((double)mtr * p->subsmpl_p);
	if (is_two_stage(p, mtr)) {
		std::vector<unsigned> cnt(M);
		return masked_two_stage(pl, p, d_x, ld, mtr, h_sel, M, d_ls_out, d_ts_out, cnt.data(), s, false, nullptr, nullptr);
	}
	const size_t N = pl->N, nc = pl->ncoef;
	int rc;
	void *v;
	if ((rc = scratch(pl, SCR_SEL, (size_t)M * mtr, &v))) return rc;
	char *d_sel = (char *)v;
	HIP_TRY(hipMemcpyAsync(d_sel, h_sel, (size_t)M * mtr, hipMemcpyHostToDevice, st));
	if ((rc = scratch(pl, SCR_SUBST, (size_t)M * nc * 2 * sizeof(double2), &v))) return rc;
	double2 *STm = (double2 *)v, *PSm = STm + (size_t)M * nc;
	HIP_TRY(hipMemsetAsync(STm, 0, (size_t)M * nc * 2 * sizeof(double2), st));
	// the traces are transformed 64 at a time (the forward kernels fill the GPU far better than with 8); the masked accumulation walks
	// a batch in trace order
	const size_t FB = std::min<size_t>(64, std::max<size_t>(8, (tspws_part_budget_bytes() / (pl->npart * sizeof(double2))) & ~(size_t)7));
	if ((rc = scratch(pl, SCR_PART, FB * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	for (size_t t0 = 0; t0 < mtr; t0 += FB) {
		const size_t nf = std::min(FB, mtr - t0);
		if ((rc = tspws_forward_parts_f32(pl, d_x + t0 * ld, nf, ld, part, st, nullptr, ScaleRange()))) return rc;
		hipLaunchKernelGGL(k_accumulate_masked, dim3(pl->acc_blocks, (M + 7) / 8), dim3(256), 0, st, (const double2 *)part, pl->npart, pl->d_sc, pl->S,
		                   nc, (unsigned)nf, d_sel, mtr, t0, M, STm, PSm);
	}
	const float scale = (float)(1. / (double)K); // fa1 = W[m]/K with W = 1 (:580)
	hipLaunchKernelGGL(k_sub_linear, dim3((unsigned)((N + 255) / 256), M), dim3(256), 0, st, d_x, ld, N, mtr, d_sel, scale, d_ls_out);
	// weights, inverses and float casts of the M subsamples side by side (every subsample has K traces)
	if ((rc = scratch(pl, SCR_JKOUT, (size_t)M * (2 * nc + N) * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *xr = OUT + (size_t)M * 2 * nc;
	tspws_weight_batched(pl, (double2 *)OUT, (const double2 *)STm, (const double2 *)PSm, tspws_weight_mode(p->wu, p->unbiased, (unsigned)K), (double)(unsigned)K,
	                     p->wu, nullptr, M, nc, nc, st, (double)(unsigned)K);
	if ((rc = tspws_hip_inverse(pl, OUT, M, xr, s))) return rc;
	tspws_epilogue_rows(d_ts_out, xr, N, M, st);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st)); // `sel` goes out of scope
	return 0;
}

// ------------------------------------------------------------------------------------------
// convergence curves (ts_pws1f_lib.c:247-314, similarity :433-449, misfit :452-462)
// ------------------------------------------------------------------------------------------
// out[0] = sum d*r, out[1] = sum d*d, out[2] = sum (d-r)^2, out[3] = sum r*r ; one workgroup, fixed order
// blockIdx.x = row of d (N apart), its four sums 4 apart in out
__global__ void __launch_bounds__(1024) k_dot4(const double *__restrict__ d, const float *__restrict__ r, size_t N, double *__restrict__ out)
{
	d += (size_t)blockIdx.x * N; out += (size_t)blockIdx.x * 4;
	__shared__ double red[16][4];
	double a = 0, b = 0, c = 0, e = 0;
	for (size_t n = threadIdx.x; n < N; n += 1024) {
		const double dv = d[n], rv = (double)r[n], df = dv - rv;
		a = fma(dv, rv, a); b = fma(dv, dv, b); c = fma(df, df, c); e = fma(rv, rv, e);
	}
	a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); e = wave_sum(e);
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; red[threadIdx.x >> 6][2] = c; red[threadIdx.x >> 6][3] = e; }
	__syncthreads();
	if (threadIdx.x < 4) {
		double t = 0;
		for (int w = 0; w < 16; w++) t += red[w][threadIdx.x];
		out[threadIdx.x] = t;
	}
}

// running linear stack: d += x_i; d *= (float)(1/(i+1)); metrics; d *= (i+1)   (:288-308, literal FLOAT reciprocal)
// every workgroup writes its partial sums of the three metrics per step; k_conv_lin_reduce adds them in order
__global__ void __launch_bounds__(256) k_conv_linear(const float *__restrict__ x, size_t ld, size_t N, size_t mtr, const float *__restrict__ ref,
                                                     double *__restrict__ partial, float *__restrict__ steps)
{
	__shared__ double red[4][3];
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	const bool live = n < N;
	const double rv = live ? (double)ref[n] : 0.0;
	double d = 0;
	for (size_t i = 0; i < mtr; i++) {
		if (live) d += (double)x[i * ld + n];
		const float inv = (float)(1.0 / (double)(i + 1));
		d *= (double)inv;
		const double df = d - rv;
		double a = live ? d * rv : 0.0, b = live ? d * d : 0.0, c = live ? df * df : 0.0;
		a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; red[threadIdx.x >> 6][2] = c; }
		__syncthreads();
		if (threadIdx.x < 3)
			partial[((size_t)blockIdx.x * mtr + i) * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
		__syncthreads();
		if (steps && live) steps[i * N + n] = (float)d;
		d *= (double)(i + 1);
	}
}

__global__ void __launch_bounds__(256) k_conv_lin_reduce(const double *__restrict__ partial, unsigned nblocks, size_t mtr, double *__restrict__ out)
{
	const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; // index into [mtr][3]
	if (j >= mtr * 3) return;
	double t = 0;
	for (unsigned b = 0; b < nblocks; b++) t += partial[(size_t)b * mtr * 3 + j];
	out[j] = t;
}

extern "C" int tspws_hip_convergence(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const float *d_ref_ts,
                                     const float *d_ref_ls, double *h_ts_sim, double *h_ts_misfit, double *h_ls_sim, double *h_ls_misfit,
                                     float *d_ts_steps, float *d_ls_steps, void *s)
{
	if (!pl || !p || !d_x || !d_ref_ts || !d_ref_ls || !h_ts_sim || !h_ts_misfit || !h_ls_sim || !h_ls_misfit)
		return fail(TSPWS_E_ARG, "convergence: NULL");
	if (!mtr) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const size_t N = pl->N, nc = pl->ncoef;
	int rc;
	void *v;
	if ((rc = scratch(pl, SCR_OUT, 6 * nc * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *ST = OUT + 2 * nc, *PS = ST + 2 * nc;
	if ((rc = scratch(pl, SCR_X2, 2 * N * sizeof(double), &v))) return rc;
	double *xr = (double *)v;
	const unsigned nblk = (unsigned)((N + 255) / 256);
	if ((rc = scratch(pl, SCR_CONV, ((size_t)mtr * 4 + (size_t)nblk * mtr * 3 + mtr * 3) * sizeof(double), &v))) return rc;
	double *d_ts = (double *)v, *d_lpart = d_ts + mtr * 4, *d_lin = d_lpart + (size_t)nblk * mtr * 3;
	// The single-stage steps (tspws_stacks_float_1step, :835-863: Tr <= Kmax or no two-stage at all) add ONE trace to the running
	// stacks each.  They are taken in batches: the traces of a batch are transformed together (the kernels fill the GPU far better
	// with 64 traces than with one), k_accumulate_parts adds them one by one in trace order and writes the weighted coefficients
	// after every trace (prefix outputs), and the batch's inverses, similarity sums and float casts run side by side.
	const size_t n_single = p->Kmax ? std::min<size_t>(mtr, p->Kmax) : mtr;
	const size_t budget = tspws_part_budget_bytes();
	const size_t per_step = (pl->npart + nc) * sizeof(double2) + N * sizeof(double);
	const size_t FB = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(64, n_single), budget / std::max<size_t>(1, per_step)));
	if ((rc = scratch(pl, SCR_PART, std::max<size_t>(2, FB) * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	if ((rc = scratch(pl, SCR_JKOUT, FB * (2 * nc + N) * sizeof(double), &v))) return rc;
	double *OUTb = (double *)v, *xrb = OUTb + FB * 2 * nc;
	double *P = nullptr;
	if (p->Kmax) { if ((rc = scratch(pl, SCR_P, (size_t)p->Kmax * N * sizeof(double), &v))) return rc; P = (double *)v; }
	for (size_t i0 = 0; i0 < n_single; i0 += FB) {
		const unsigned nb = (unsigned)std::min(FB, n_single - i0);
		if ((rc = tspws_forward_parts_f32(pl, d_x + i0 * ld, nb, ld, part, st, nullptr, ScaleRange()))) return rc;
		WeightArgs wa;
		wa.OUTP = (double2 *)OUTb; wa.outp_stride = nc; wa.k0 = (unsigned)i0; wa.wu = p->wu;
		wa.mode = tspws_weight_mode(p->wu, p->unbiased, 2); wa.mode1 = tspws_weight_mode(p->wu, p->unbiased, 1);
		tspws_launch_accumulate(pl, (const double2 *)part, nb, (double2 *)ST, (double2 *)PS, i0 == 0 ? 1 : 0, nullptr, 0, st, 1, 0, 0, nullptr, &wa, ScaleRange());
		if ((rc = tspws_hip_inverse(pl, OUTb, nb, xrb, s))) return rc;
		hipLaunchKernelGGL(k_dot4, dim3(nb), dim3(1024), 0, st, (const double *)xrb, d_ref_ts, N, d_ts + i0 * 4);
		if (d_ts_steps) tspws_epilogue_rows(d_ts_steps + i0 * N, xrb, N, nb, st);
	}
	for (size_t i = n_single; i < mtr; i++) { // two-stage over the first Tr traces, recomputed from scratch like the reference (:266-268)
		const size_t Tr = i + 1;
		const unsigned K = p->Kmax;
		if ((rc = tspws_hip_partial_stacks(pl, d_x, ld, Tr, 0, Tr, K, P, N, s))) return rc;
		if ((rc = tspws_hip_stacks_double(pl, P, K, N, ST, PS, s))) return rc;
		if ((rc = tspws_hip_weight(pl, OUT, ST, PS, K, (unsigned)Tr, p->wu, p->unbiased, s))) return rc;
		if ((rc = tspws_hip_inverse(pl, OUT, 1, xr, s))) return rc;
		hipLaunchKernelGGL(k_dot4, dim3(1), dim3(1024), 0, st, (const double *)xr, d_ref_ts, N, d_ts + i * 4);
		if (d_ts_steps && (rc = tspws_hip_epilogue(nullptr, d_ts_steps + i * N, nullptr, xr, N, 1, s))) return rc;
	}
	hipLaunchKernelGGL(k_conv_linear, dim3(nblk), dim3(256), 0, st, d_x, ld, N, mtr, d_ref_ls, d_lpart, d_ls_steps);
	hipLaunchKernelGGL(k_conv_lin_reduce, dim3((unsigned)((mtr * 3 + 255) / 256)), dim3(256), 0, st, (const double *)d_lpart, nblk, mtr, d_lin);
	hipLaunchKernelGGL(k_dot4, dim3(1), dim3(1024), 0, st, (const double *)xr, d_ref_ls, N, d_lpart); // only out[3] = sum ref_ls^2 is used
	HIP_TRY(hipGetLastError());
	std::vector<double> hts(mtr * 4), hl(mtr * 3);
	double lsq[4];
	HIP_TRY(hipMemcpyAsync(hts.data(), d_ts, mtr * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hl.data(), d_lin, mtr * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(lsq, d_lpart, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (size_t i = 0; i < mtr; i++) {
		h_ts_sim[i] = hts[i * 4] / sqrt(hts[i * 4 + 1]) / sqrt(hts[i * 4 + 3]);
		h_ts_misfit[i] = hts[i * 4 + 2];
		h_ls_sim[i] = hl[i * 3] / sqrt(hl[i * 3 + 1]) / sqrt(lsq[3]);
		h_ls_misfit[i] = hl[i * 3 + 2];
	}
	return 0;
}
