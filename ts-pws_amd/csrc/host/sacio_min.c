/* sacio_min.c -- see sacio_min.h */
#include "sacio_min.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }

static void swap_words(void *p, size_t nwords)
{
	uint32_t *w = (uint32_t *)p;
	for (size_t i = 0; i < nwords; i++) w[i] = bswap32(w[i]);
}

int sac_read(const char *path, sac_header *h, float *data, int maxpts, int *npts_read)
{
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	if (fread(h, 1, sizeof *h, f) != sizeof *h) { fclose(f); return -2; }
	int swapped = 0;
	int32_t ver = h->i[SAC_I_NVHDR];
	if (ver < 1 || ver > 7) {
		ver = (int32_t)bswap32((uint32_t)ver);
		if (ver < 1 || ver > 7) { fclose(f); return -3; }
		swapped = 1;
		swap_words(h->f, 70);
		swap_words(h->i, 40);
	}
	const int npts = h->i[SAC_I_NPTS];
	if (npts_read) *npts_read = npts;
	if (data) {
		if (npts < 0) { fclose(f); return -3; }
		const int n = npts < maxpts ? npts : maxpts;
		if ((int)fread(data, sizeof(float), (size_t)n, f) != n) { fclose(f); return -4; }
		if (swapped) swap_words(data, (size_t)n);
	}
	fclose(f);
	return 0;
}

void sac_new_header(sac_header *h)
{
	for (int i = 0; i < 70; i++) h->f[i] = SAC_UNDEF_F;
	for (int i = 0; i < 40; i++) h->i[i] = SAC_UNDEF_I;
	for (int i = 0; i < 192; i += 8) memcpy(h->k + i, "-12345  ", 8);
	memcpy(h->k + SAC_K_KEVNM, "-12345          ", 16);
	h->i[SAC_I_NVHDR] = 6;
	h->i[SAC_I_IFTYPE] = 1;  /* ITIME */
	h->i[SAC_I_LEVEN] = 1;
	h->i[SAC_I_LPSPOL] = 0;
	h->i[SAC_I_LOVROK] = 1;
	h->i[SAC_I_LCALDA] = 1;
	h->i[39] = 0;
}

void sac_set_k(sac_header *h, int off, int len, const char *s)
{
	size_t n = strlen(s);
	if (n > (size_t)len) n = (size_t)len;
	memset(h->k + off, ' ', (size_t)len);
	memcpy(h->k + off, s, n);
}

void sac_get_k(const sac_header *h, int off, int len, char *dst)
{
	memcpy(dst, h->k + off, (size_t)len);
	dst[len] = '\0';
	for (int i = len - 1; i >= 0 && (dst[i] == ' ' || dst[i] == '\0'); i--) dst[i] = '\0';
	if (!strncmp(dst, "-12345", 6)) dst[0] = '\0';
}

int sac_write(const char *path, sac_header *h, const float *data)
{
	const int n = h->i[SAC_I_NPTS];
	if (n > 0) {
		float mn = data[0], mx = data[0];
		double sum = 0;
		for (int i = 0; i < n; i++) { if (data[i] < mn) mn = data[i]; if (data[i] > mx) mx = data[i]; sum += data[i]; }
		h->f[SAC_F_DEPMIN] = mn; h->f[SAC_F_DEPMAX] = mx; h->f[SAC_F_DEPMEN] = (float)(sum / n);
		h->f[SAC_F_E] = h->f[SAC_F_B] + (float)(n - 1) * h->f[SAC_F_DELTA];
	}
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	int ok = fwrite(h, 1, sizeof *h, f) == sizeof *h;
	if (ok && n > 0) ok = (int)fwrite(data, sizeof(float), (size_t)n, f) == n;
	fclose(f);
	return ok ? 0 : -2;
}

/* days since 1970-01-01 of Jan 1st of `year` (proleptic Gregorian) */
static long days_to_year(long y)
{
	y -= 1;
	return y * 365 + y / 4 - y / 100 + y / 400 - 719162L; /* 719162 = days of years 1..1969 */
}

time_t sac_reference_time(const sac_header *h)
{
	const int32_t *i = h->i;
	if (i[SAC_I_NZYEAR] == SAC_UNDEF_I || i[SAC_I_NZJDAY] == SAC_UNDEF_I) return 0;
	long d = days_to_year(i[SAC_I_NZYEAR]) + (i[SAC_I_NZJDAY] - 1);
	long s = 0;
	if (i[SAC_I_NZHOUR] != SAC_UNDEF_I) s += 3600L * i[SAC_I_NZHOUR];
	if (i[SAC_I_NZMIN] != SAC_UNDEF_I) s += 60L * i[SAC_I_NZMIN];
	if (i[SAC_I_NZSEC] != SAC_UNDEF_I) s += i[SAC_I_NZSEC];
	return (time_t)(d * 86400L + s);
}
