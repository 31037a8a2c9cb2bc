/*
 * oracle/orc_io.c -- TEST INFRASTRUCTURE (see orc.h).
 * Big-endian "machine independent" model files:
 *   btk/common/mach_ind_io.cc:176-200,331-350,409-430 read/write int, float, short (big endian)
 *   btk/common/mach_ind_io.cc:490-510,1009-1022        strings: int16 length + len+1 bytes (NUL)
 *   asr/gaussian/codebookBasic.cc:258-309,352-405       CodebookBasic::load / save
 *   asr/gaussian/codebookBasic.cc:906-983               CodebookSetBasic::load / save
 *   asr/gaussian/distribBasic.cc:85-142,254-304         DistribBasic / DistribSetBasic load / save
 */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CODEBOOK_MAGIC 64207531
#define MARKER_MAGIC   123456789
#define COV_DIAGONAL   2

static int rd_i32(FILE* fp) { unsigned char b[4] = {0,0,0,0}; if (fread(b, 1, 4, fp) != 4) return 0; return (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); }
static float rd_f32(FILE* fp) { int i = rd_i32(fp); float f; memcpy(&f, &i, 4); return f; }
static short rd_i16(FILE* fp) { unsigned char b[2] = {0,0}; if (fread(b, 1, 2, fp) != 2) return 0; return (short) ((unsigned) b[0] << 8 | (unsigned) b[1]); }
static void wr_i32(FILE* fp, int v) { unsigned u = (unsigned) v; unsigned char b[4] = { (unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); }
static void wr_f32(FILE* fp, float f) { int i; memcpy(&i, &f, 4); wr_i32(fp, i); }
static void wr_i16(FILE* fp, short v) { unsigned short u = (unsigned short) v; unsigned char b[2] = { (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 2, fp); }
static void wr_str(FILE* fp, const char* s) { short len = (short) strlen(s); wr_i16(fp, len); fwrite(s, (size_t) len + 1, 1, fp); }
static char* rd_str(FILE* fp) { short len = rd_i16(fp); char* s = (char*) calloc((size_t) len + 2, 1); if (fread(s, (size_t) len + 1, 1, fp) != 1) { s[0] = 0; } return s; }

int orc_cbset_save(const orc_cbset* cb, const char** names, const char* file)
{
  FILE* fp = fopen(file, "wb"); if (!fp) return -7;
  wr_i32(fp, CODEBOOK_MAGIC); wr_i32(fp, 0); wr_i32(fp, cb->K);
  for (int k = 0; k < cb->K; k++) {
    wr_str(fp, names[k]);
    wr_i32(fp, cb->refN[k]); wr_i32(fp, cb->dimN); wr_i32(fp, cb->dimN); wr_i32(fp, 1);
    wr_i32(fp, COV_DIAGONAL); wr_i32(fp, 0); wr_i32(fp, 0);
    for (int i = 0; i < cb->refN[k]; i++) {
      int g = cb->off[k] + i;
      wr_f32(fp, cb->count[g]);
      for (int d = 0; d < cb->dimN; d++) wr_f32(fp, cb->mean[(size_t) g * cb->dimN + d]);
      for (int d = 0; d < cb->dimN; d++) wr_f32(fp, cb->ivar[(size_t) g * cb->dimN + d]);
      wr_f32(fp, cb->det[g]);
    }
    wr_i32(fp, MARKER_MAGIC);
  }
  fclose(fp);
  return 0;
}

orc_cbset* orc_cbset_load(const char* file, char*** namesOut)
{
  FILE* fp = fopen(file, "rb"); if (!fp) return NULL;
  if (rd_i32(fp) != CODEBOOK_MAGIC) { fclose(fp); return NULL; }
  int cb0 = rd_i32(fp), cbN = rd_i32(fp); int K = cbN - cb0;
  orc_cbset* cb = (orc_cbset*) calloc(1, sizeof(orc_cbset));
  cb->K = K; cb->refN = (int*) calloc((size_t) K, sizeof(int)); cb->off = (int*) calloc((size_t) K + 1, sizeof(int));
  cb->pi = (float*) calloc((size_t) K, sizeof(float)); cb->scale = (float*) calloc((size_t) K, sizeof(float));
  char** names = (char**) calloc((size_t) K, sizeof(char*));
  size_t capG = 0;
  for (int k = 0; k < K; k++) {
    names[k] = rd_str(fp);
    int refN = rd_i32(fp), dimN = rd_i32(fp), orgDimN = rd_i32(fp), nSub = rd_i32(fp), ctype = rd_i32(fp);
    int regP = rd_i32(fp), descP = rd_i32(fp);
    (void) ctype;
    cb->refN[k] = refN; cb->dimN = dimN; cb->off[k + 1] = cb->off[k] + refN;
    cb->pi[k] = log(2.0 * M_PI) * dimN;   /* codebookBasic.cc:170,283 (float _pi) */
    cb->scale[k] = 1.0f;
    size_t G = (size_t) cb->off[k + 1];
    if (G > capG) {
      capG = G * 2;
      cb->mean = (float*) realloc(cb->mean, sizeof(float) * capG * dimN); cb->ivar = (float*) realloc(cb->ivar, sizeof(float) * capG * dimN);
      cb->det = (float*) realloc(cb->det, sizeof(float) * capG); cb->count = (float*) realloc(cb->count, sizeof(float) * capG);
    }
    for (int i = 0; i < refN; i++) {
      int g = cb->off[k] + i;
      cb->count[g] = rd_f32(fp);
      for (int d = 0; d < orgDimN; d++) { float v = rd_f32(fp); if (d < dimN) cb->mean[(size_t) g * dimN + d] = v; }
      for (int d = 0; d < dimN; d++) cb->ivar[(size_t) g * dimN + d] = rd_f32(fp);
      cb->det[g] = rd_f32(fp);
    }
    if (regP) for (int i = 0; i < refN; i++) { int n = (unsigned short) rd_i16(fp); for (int c = 0; c < n; c++) rd_i16(fp); }
    if (descP) for (int i = 0; i < refN; i++) for (int b = 0; b < nSub; b++) rd_i16(fp);
    if (rd_i32(fp) != MARKER_MAGIC) { fclose(fp); orc_cbset_free(cb); return NULL; }
  }
  fclose(fp);
  if (namesOut) *namesOut = names; else { for (int k = 0; k < K; k++) free(names[k]); free(names); }
  return cb;
}

void orc_cbset_free(orc_cbset* cb)
{
  if (!cb) return;
  free(cb->refN); free(cb->off); free(cb->mean); free(cb->ivar); free(cb->det); free(cb->count); free(cb->pi); free(cb->scale); free(cb);
}

int orc_distset_save(int n, const char** names, const char** cbnames, const int* refN,
                     const float* count, const float* const* val, const char* file)
{
  FILE* fp = fopen(file, "wb"); if (!fp) return -7;
  wr_i32(fp, n);
  for (int i = 0; i < n; i++) {
    wr_str(fp, names[i]); wr_str(fp, cbnames[i]);
    wr_i32(fp, -refN[i]); wr_f32(fp, count[i]);
    for (int r = 0; r < refN[i]; r++) wr_f32(fp, val[i][r]);
  }
  fclose(fp);
  return 0;
}

int orc_distset_load(const char* file, int* nOut, char*** names, char*** cbnames, int** refN,
                     float** count, float*** val)
{
  FILE* fp = fopen(file, "rb"); if (!fp) return -7;
  int n = rd_i32(fp);
  *nOut = n; *names = (char**) calloc((size_t) n, sizeof(char*)); *cbnames = (char**) calloc((size_t) n, sizeof(char*));
  *refN = (int*) calloc((size_t) n, sizeof(int)); *count = (float*) calloc((size_t) n, sizeof(float)); *val = (float**) calloc((size_t) n, sizeof(float*));
  for (int i = 0; i < n; i++) {
    (*names)[i] = rd_str(fp); (*cbnames)[i] = rd_str(fp);
    int r = rd_i32(fp);
    if (r < 0) { r = -r; (*count)[i] = rd_f32(fp); }
    (*refN)[i] = r; (*val)[i] = (float*) calloc((size_t) r, sizeof(float));
    for (int j = 0; j < r; j++) (*val)[i][j] = rd_f32(fp);
  }
  fclose(fp);
  return 0;
}
