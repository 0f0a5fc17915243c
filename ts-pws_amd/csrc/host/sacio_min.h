/*
 * sacio_min.h -- minimal SAC (binary, header version 6/7) and msacs reader/writer for the ts_pws front-end.
 *
 * The reference front-end uses IRIS SAC's sacio.a (rsach/rsac1/get?hv/set?hv/wsac0,
 * /root/reference/src/ts_pws1f.c:617-636,684,733-758), which is not vendored and not available
 * here, so this file implements just the subset of the published SAC binary format those calls
 * touch: the 632-byte header (70 float, 40 int, 192 char words) followed by npts float32 samples,
 * either byte order.  The msacs container is the reference's own format (src/sac2bin.h:6-27,
 * sac2bin.c:192-195): 116-byte header, time_t[nseq], float lag0[nseq], float data[nseq][nlags].
 */
#ifndef SACIO_MIN_H
#define SACIO_MIN_H

#include <stdint.h>
#include <time.h>

#define SAC_UNDEF_F (-12345.0f)
#define SAC_UNDEF_I (-12345)

typedef struct {
	float    f[70];   /* delta=0 b=5 e=6 stla=31 stlo=32 stel=33 evla=35 evlo=36 user0=40 depmin=1 depmax=2 depmen=56 */
	int32_t  i[40];   /* nzyear=0 nzjday=1 nzhour=2 nzmin=3 nzsec=4 nzmsec=5 nvhdr=6 npts=9 iftype=15 iztype=17 leven=35 lovrok=37 lcalda=38 */
	char     k[192];  /* kstnm@0 kevnm@8(16) khole@24 kuser0@136 kuser1@144 kuser2@152 kcmpnm@160 knetwk@168 kinst@184 */
} sac_header;

enum { SAC_F_DELTA = 0, SAC_F_DEPMIN = 1, SAC_F_DEPMAX = 2, SAC_F_B = 5, SAC_F_E = 6, SAC_F_STLA = 31, SAC_F_STLO = 32,
       SAC_F_STEL = 33, SAC_F_EVLA = 35, SAC_F_EVLO = 36, SAC_F_USER0 = 40, SAC_F_DEPMEN = 56 };
enum { SAC_I_NZYEAR = 0, SAC_I_NZJDAY = 1, SAC_I_NZHOUR = 2, SAC_I_NZMIN = 3, SAC_I_NZSEC = 4, SAC_I_NZMSEC = 5, SAC_I_NVHDR = 6,
       SAC_I_NPTS = 9, SAC_I_IFTYPE = 15, SAC_I_IZTYPE = 17, SAC_I_LEVEN = 35, SAC_I_LPSPOL = 36, SAC_I_LOVROK = 37, SAC_I_LCALDA = 38 };
enum { SAC_K_KSTNM = 0, SAC_K_KEVNM = 8, SAC_K_KHOLE = 24, SAC_K_KUSER0 = 136, SAC_K_KUSER1 = 144, SAC_K_KUSER2 = 152,
       SAC_K_KCMPNM = 160, SAC_K_KNETWK = 168, SAC_K_KINST = 184 };

/* Read header (and, when data != NULL, up to maxpts samples; *npts_read = samples in the file).
 * Returns 0, or <0: -1 open, -2 short header, -3 not a SAC file, -4 short data. */
int  sac_read(const char *path, sac_header *h, float *data, int maxpts, int *npts_read);
void sac_new_header(sac_header *h);                       /* all fields undefined, nvhdr 6, evenly spaced time series */
void sac_set_k(sac_header *h, int off, int len, const char *s);   /* blank-padded, like setkhv */
void sac_get_k(const sac_header *h, int off, int len, char *dst); /* NUL-terminated, trailing blanks stripped, "" if undefined */
int  sac_write(const char *path, sac_header *h, const float *data); /* fills depmin/depmax/depmen/e; native byte order */
time_t sac_reference_time(const sac_header *h);           /* UTC of the header's reference time, 0 when undefined */

/* msacs container header, byte-compatible with the reference's t_ccheader (src/sac2bin.h:6-27) */
typedef struct {
	char     method[8], net1[8], sta1[8], loc1[8], chn1[8], net2[8], sta2[8], loc2[8], chn2[8];
	float    stlat1, stlon1, stel1, stlat2, stlon2, stel2;
	uint32_t nlags, nseq;
	float    tlength, lag1, lag2;
} msacs_header;

#endif
