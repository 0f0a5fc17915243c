/*
 * ts_pws1f_lib.h -- drop-in boundary of the MI355X ts-PWS stacking engine.
 *
 * This header re-declares, in this project's own words, the caller-visible ABI
 * of the reference stacking library so that a program written against
 *   /root/reference/src/ts_pws1f_lib.h:19-57   (t_tsPWS)
 *   /root/reference/src/ts_pws1f_lib.h:59-77   (t_hdr)
 *   /root/reference/src/ts_pws1f_lib.h:79-94   (t_tsPWS_out)
 *   /root/reference/src/ts_pws1f_lib.h:96-102  (t_data)
 *   /root/reference/src/ts_pws1f_lib.h:104     (tspws_main)
 * links against libtspws_hip.so unchanged.  Both reference callers use
 * positional initialisers (ts_pws1f.c:140-142, gw_ts_pws.c:15-16), so member
 * ORDER and TYPES are ABI; the static assertions at the bottom pin the x86-64
 * offsets recorded in SURVEY.md section 8(b).
 *
 * Ownership (same contract as the reference, ts_pws1f.c:229-268): the caller
 * allocates and frees every buffer reachable from these structs; the library
 * only owns scratch it frees before returning.
 */
#ifndef TSPWS_LIB
#define TSPWS_LIB

#include <stddef.h>
#include <time.h>

#ifndef PI
#define PI 3.14159265358979328
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stacking request ------------------------------------------------------
 * Fields the engine may rewrite in place (the reference does the same at
 * ts_pws1f_lib.c:74,93,96-123): fold, fmin, w0, V, b0, s0, J.               */
typedef struct {
	int          type;         /* -1 Morlet, -2 exact (zero-mean) Morlet, -3 complex Mexican hat */
	unsigned int uni;          /* 1: every scale sampled at rate 1 (no decimation)                */
	unsigned int J;            /* octaves; 0 = derive from the trace length / fmin                */
	unsigned int V;            /* voices per octave                                               */
	double       s0;           /* finest scale                                                    */
	double       b0;           /* translation step at scale 1                                     */
	double       w0;           /* Morlet centre frequency                                         */
	double       wu;           /* phase-weight exponent                                           */
	double       fmin;         /* lowest analysed frequency (Hz); 0 = automatic                   */
	double       Q;            /* Morlet shape given as quality factor  (used when w0set == 1)    */
	double       cycle;        /* Morlet shape given as cycle count     (used when w0set == 2)    */
	int          w0set;        /* which of Q / cycle / w0 the caller supplied                     */
	int          lrm;          /* subtract each trace's mean first                                */
	int          bin;          /* front-end only: input is an msacs binary                        */
	int          lkinst;       /* front-end only                                                  */
	int          lVfix;        /* caller fixed V  (suppresses auto-derivation)                    */
	int          ls0fix;       /* caller fixed s0                                                 */
	int          lb0fix;       /* caller fixed b0                                                 */
	int          verbose;
	int          fold;         /* average causal and acausal lags in place                        */
	int          unbiased;     /* bias-corrected coherence (honoured only when wu == 2)           */
	int          convergence;  /* fill the *_sim / *_misfit curves                                */
	unsigned int subsmpl_N;    /* random subsampling: realisations                                */
	double       subsmpl_p;    /* random subsampling: kept fraction                               */
	unsigned int jackknife_n;  /* jackknife: day-of-year bins                                     */
	unsigned int jackknife_d;  /* jackknife: bins deleted per replica                             */
	unsigned int obin;         /* front-end only                                                  */
	int          AllSteps;     /* with convergence: keep the stack after every added trace        */
	unsigned int Nmax;         /* use only the first Nmax traces (0 = all)                        */
	unsigned int Kmax;         /* two-stage: number of partial linear stacks (0 = single stage)   */
	char        *kinst;        /* front-end only                                                  */
	char        *filein;       /* front-end only                                                  */
	char        *fileout;      /* front-end only                                                  */
	char        *fileconv;     /* front-end only                                                  */
} t_tsPWS;

/* ---- common trace header --------------------------------------------------- */
typedef struct {
	int          max;          /* samples per trace */
	unsigned int mtr;          /* traces            */
	float        evla, evlo;   /* station-1 ("event") coordinates */
	float        stla, stlo;   /* station-2 coordinates           */
	float        stel;
	float        dt;           /* sampling interval (s)  */
	float        beg;          /* time of sample 0 (s)   */
	char         net1[9], sta1[9], loc1[9], chn1[9];
	char         net2[9], sta2[9], loc2[9], chn2[9];
} t_hdr;

/* ---- results (all caller-allocated) ---------------------------------------- */
typedef struct {
	float        *ls;             /* [max] frame-filtered linear stack                    */
	float        *tsPWS;          /* [max] time-scale phase-weighted stack                */
	double       *ls_sim;         /* [mtr] convergence curves ...                         */
	double       *tsPWS_sim;
	double       *ls_misfit;
	double       *tsPWS_misfit;
	float        *ls_steps;       /* [mtr][max] or NULL                                   */
	float        *tsPWS_steps;
	float       **ls_subsmpl;     /* [M] row pointers into one M*max block                */
	float       **tsPWS_subsmpl;
	unsigned int *mtr_subsmpl;    /* [M] traces used by each replica                      */
	unsigned int  M;              /* replicas (C(n,d) for the jackknife)                  */
	unsigned int  N;
	unsigned int  mtr;
} t_tsPWS_out;

/* ---- input ensemble --------------------------------------------------------- */
typedef struct {
	float  *sigall;     /* [mtr][max] row-major, one trace per row; modified by fold / rm */
	time_t *time;       /* [mtr] start times (needed by the jackknife only)               */
	float  *lag0;       /* [mtr] front-end only                                           */
	t_hdr   hdr;
	float  *reference;  /* optional [max] reference trace for the convergence curves      */
} t_data;

/* Replaces /root/reference/src/ts_pws1f_lib.h:104 (body ts_pws1f_lib.c:48-352).
 * Returns 0 on success, -1 for a NULL argument, 4 when scratch could not be
 * obtained (host or device), exactly like the reference; additionally 5 when
 * no HIP device / kernel image is usable (never silently falls back to a CPU). */
int tspws_main(t_tsPWS *tspws, t_tsPWS_out *out, t_data *in);

#ifdef __cplusplus
}
#endif

/* ABI pins (x86-64) */
#if defined(__x86_64__) && (defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L || defined(__cplusplus))
#ifdef __cplusplus
#define TSPWS_SA(c, m) static_assert(c, m)
#else
#define TSPWS_SA(c, m) _Static_assert(c, m)
#endif
TSPWS_SA(sizeof(t_tsPWS) == 184, "t_tsPWS size");
TSPWS_SA(offsetof(t_tsPWS, s0) == 16 && offsetof(t_tsPWS, w0) == 32 && offsetof(t_tsPWS, wu) == 40, "t_tsPWS head");
TSPWS_SA(offsetof(t_tsPWS, cycle) == 64 && offsetof(t_tsPWS, w0set) == 72, "t_tsPWS shape");
TSPWS_SA(offsetof(t_tsPWS, unbiased) == 108 && offsetof(t_tsPWS, subsmpl_N) == 116, "t_tsPWS flags");
TSPWS_SA(offsetof(t_tsPWS, subsmpl_p) == 120 && offsetof(t_tsPWS, jackknife_n) == 128, "t_tsPWS resampling");
TSPWS_SA(offsetof(t_tsPWS, Nmax) == 144 && offsetof(t_tsPWS, Kmax) == 148, "t_tsPWS limits");
TSPWS_SA(offsetof(t_tsPWS, kinst) == 152 && offsetof(t_tsPWS, fileconv) == 176, "t_tsPWS strings");
TSPWS_SA(sizeof(t_hdr) == 108, "t_hdr size");
TSPWS_SA(sizeof(t_tsPWS_out) == 104 && offsetof(t_tsPWS_out, ls_subsmpl) == 64, "t_tsPWS_out");
TSPWS_SA(offsetof(t_tsPWS_out, mtr_subsmpl) == 80 && offsetof(t_tsPWS_out, M) == 88, "t_tsPWS_out tail");
TSPWS_SA(offsetof(t_tsPWS_out, N) == 92 && offsetof(t_tsPWS_out, mtr) == 96, "t_tsPWS_out tail2");
TSPWS_SA(sizeof(t_data) == 144 && offsetof(t_data, hdr) == 24 && offsetof(t_data, reference) == 136, "t_data");
#undef TSPWS_SA
#endif

#endif /* TSPWS_LIB */
