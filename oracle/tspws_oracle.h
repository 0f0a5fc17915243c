/*
 * tspws_oracle.h -- TEST INFRASTRUCTURE ONLY (see tspws_oracle.c).
 * CPU restatement of the reference ts-PWS path; used by tests/, smoke() and
 * bench.py's cpu_baseline leg as the checker.  Never linked into the product.
 */
#ifndef TSPWS_ORACLE_H
#define TSPWS_ORACLE_H

#include <stddef.h>
#include <time.h>
#include "../include/ts_pws1f_lib.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_frame orc_frame;

orc_frame *orc_frame_create(int type, unsigned J, unsigned V, unsigned N, double s0, double b0, double w0, int uni);
void       orc_frame_destroy(orc_frame *f);
unsigned   orc_frame_S(const orc_frame *f);
size_t     orc_frame_ncoef(const orc_frame *f);
size_t     orc_frame_ntaps(const orc_frame *f);
double     orc_frame_cpsi(const orc_frame *f);
void       orc_frame_tables(const orc_frame *f, double *scale, unsigned *L, int *c, int *cd, unsigned *D, unsigned *Ns);
void       orc_frame_taps(const orc_frame *f, double *w, double *wd); /* interleaved re,im */

void orc_forward(const orc_frame *f, const double *x, double *Y);      /* Y: 2*ncoef doubles */
void orc_inverse(const orc_frame *f, const double *Y, double *xrec);
void orc_accumulate(double *ST, double *PS, const double *Y, size_t ncoef);
void orc_weight(double *OUT, const double *ST, const double *PS, size_t ncoef, unsigned K, unsigned M, double wu, int unbiased);
void orc_partial_stacks(double *P, const float *sigall, size_t max, size_t mtr, unsigned Kmax);
void orc_resolve(t_tsPWS *p, unsigned nsamp, float dt);
int  orc_subsampling_plan(char *sel, size_t J, size_t K);
int  orc_jackknife_plan(char *sel, const time_t *tm, size_t mtr, unsigned d, unsigned n, unsigned C);
int  orc_tspws_main(t_tsPWS *p, t_tsPWS_out *out, t_data *in);
int  orc_tspws_main_mt(t_tsPWS *p, t_tsPWS_out *out, t_data *in); /* OpenMP trace-/scale-parallel variant, bit-identical results */

#ifdef __cplusplus
}
#endif
#endif
