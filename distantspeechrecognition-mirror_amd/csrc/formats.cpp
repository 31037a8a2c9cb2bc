// csrc/formats.cpp -- on-disk formats either side of the hot path, host side:
//   Janus "FMAT" / "FVEC" matrices and vectors and GSL raw blocks   gsl_matrix_float_load / gsl_vector_float_load (btk/matrix/gslmatrix.cc:27-96,133-240)
//   HTK parameter files                                              ReadHTKHeader / WriteHTKHeader / Read|WriteFloatBinary, WriteHTKFeatureFile,
//                                                                    HTKFeature (btk/feature/feature.cc:4025-4318)
// Big-endian primitives as btk/common/mach_ind_io.cc:176-510 (ints and floats byte-reversed on a little-endian host).
#include "common.h"
#include <sys/stat.h>

using namespace dsr;

namespace {
struct F {
  FILE* fp;
  explicit F(const char* name, const char* mode) : fp(fopen(name, mode)) { if (!fp) throw Error(DSR_E_IO, "Could not open file %s", name); }
  ~F() { if (fp) fclose(fp); }
  long left() { const long pos = ftell(fp); fseek(fp, 0, SEEK_END); const long l = ftell(fp) - pos; fseek(fp, pos, SEEK_SET); return l; }
  bool be32(int& v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) return false; v = (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); return true; }
  void wbe32(int v) { const unsigned u = (unsigned) v; const unsigned char b[4] = { (unsigned char) (u >> 24), (unsigned char) (u >> 16), (unsigned char) (u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); }
  size_t befloats(float* d, size_t n) { size_t k = 0; for (; k < n; k++) { int i; if (!be32(i)) break; memcpy(&d[k], &i, 4); } return k; }
  void wbefloats(const float* d, size_t n) { for (size_t k = 0; k < n; k++) { int i; memcpy(&i, &d[k], 4); wbe32(i); } }
};
}

extern "C" {

// gsl_matrix_float_load(m, fileName, old) into a matrix that holds rows x cols floats (its allocated size, as the reference's m).
//   old == 0: gsl_matrix_float_fread -- rows*cols native-endian floats straight from the file (gslmatrix.cc:27-38); a short file is DSR_E_IO.
//   old != 0: Janus FMAT (gslmatrix.cc:40-96): "FMAT", big-endian rows, cols, a float count that is skipped, big-endian floats.  rows < 0 in the
//             file = "number of rows wasn't set": derived from the bytes left.  Empty body: "File empty, matrix unchanged!"; a body that does not
//             match rows*cols*4 bytes: "Number of bytes in file = don't match matrix dimension"; a matrix larger than the caller's in either
//             dimension: jdimension_error "Cannot resize" (gslmatrix.cc:6-15 only ever shrinks).  *rowsOut x *colsOut is the size after the load
//             (the reference's matrix is resized to the file's; data lands row by row with the FILE's column count).
dsr_status dsr_fmat_load(const char* fileName, int old, int rows, int cols, float* data, int* rowsOut, int* colsOut)
{
  return guard([&] {
    if (!fileName || !data || rows < 1 || cols < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "rb");
    if (!old) {
      if (fread(data, sizeof(float), (size_t) rows * cols, f.fp) != (size_t) rows * cols) throw Error(DSR_E_IO, "Could not read matrix from %s", fileName);
      if (rowsOut) *rowsOut = rows; if (colsOut) *colsOut = cols;
      return;
    }
    char magic[5] = {0};
    if (fread(magic, 1, 4, f.fp) != 4 || strncmp("FMAT", magic, 4)) throw Error(DSR_E_IO, "Couldn't find magic number in file\n");
    int size1 = 0, size2 = 0, cnt = 0;
    if (!f.be32(size1) || !f.be32(size2) || !f.be32(cnt)) throw Error(DSR_E_IO, "Couldn't read the matrix header of %s", fileName);
    const long left = f.left();
    if (size2 && size1 < 0) size1 = (int) (left / (long) (sizeof(float) * (size_t) size2));      // number of rows wasn't set
    const long total = (long) size1 * (long) size2, totalBytes = total * (long) sizeof(float);
    if (!left) throw Error(DSR_E_IO, "File empty, matrix unchanged!\n");
    if (left < 0 || left != totalBytes) throw Error(DSR_E_IO, "Number of bytes in file = don't match matrix dimension:\n");
    if (size1 < 1 || size2 < 1) throw Error(DSR_E_DIMENSION, "Could not resize matrix to %d x %d.", size1, size2);
    if (rows < size1) throw Error(DSR_E_DIMENSION, "Cannot resize from %d to %d", rows, size1);
    if (cols < size2) throw Error(DSR_E_DIMENSION, "Cannot resize from %d to %d", cols, size2);
    if (f.befloats(data, (size_t) total) != (size_t) total) throw Error(DSR_E_IO, "Could not read %ld floats from %s.", total, fileName);
    if (rowsOut) *rowsOut = size1; if (colsOut) *colsOut = size2;
  });
}
// writers of the two forms (the reference writes GSL blocks with gsl_matrix_float_fwrite; FMAT files come from Janus -- written here so that
// models can be handed back): old != 0: "FMAT" + big-endian rows (or -1 when rowsUnset) / cols / count 0 + big-endian floats
dsr_status dsr_fmat_save(const char* fileName, int old, int rows, int cols, const float* data, int rowsUnset)
{
  return guard([&] {
    if (!fileName || !data || rows < 1 || cols < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "wb");
    if (!old) { if (fwrite(data, sizeof(float), (size_t) rows * cols, f.fp) != (size_t) rows * cols) throw Error(DSR_E_IO, "Could not write matrix to %s", fileName); return; }
    fwrite("FMAT", 1, 4, f.fp); f.wbe32(rowsUnset ? -1 : rows); f.wbe32(cols); { const float z = 0.0f; f.wbefloats(&z, 1); }
    f.wbefloats(data, (size_t) rows * cols);
  });
}
// gsl_vector_float_load(v, fileName, old) (gslmatrix.cc:133-240): raw block, or "FVEC" + big-endian n + count + floats; the body must hold exactly
// n floats (j_error otherwise), a vector longer than the caller's is jdimension_error "Cannot resize"
dsr_status dsr_fvec_load(const char* fileName, int old, int n, float* data, int* nOut)
{
  return guard([&] {
    if (!fileName || !data || n < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "rb");
    if (!old) { if (fread(data, sizeof(float), (size_t) n, f.fp) != (size_t) n) throw Error(DSR_E_IO, "Could not read matrix from %s", fileName); if (nOut) *nOut = n; return; }
    char magic[5] = {0};
    if (fread(magic, 1, 4, f.fp) != 4 || strncmp("FVEC", magic, 4)) throw Error(DSR_E_IO, "Couldn't find magic number in file\n");
    int m = 0, cnt = 0;
    if (!f.be32(m) || !f.be32(cnt)) throw Error(DSR_E_IO, "Couldn't read the vector header of %s", fileName);
    const long left = f.left();
    if (left < 0 || left != (long) m * (long) sizeof(float)) throw Error(DSR_E_ERROR, "Number of bytes in file = don't match vector dimension:\n");
    if (n < m) throw Error(DSR_E_DIMENSION, "Cannot resize from %d to %d", n, m);
    if (f.befloats(data, (size_t) m) != (size_t) m) throw Error(DSR_E_IO, "Could not read %d floats from %s.", m, fileName);
    if (nOut) *nOut = m;
  });
}
dsr_status dsr_fvec_save(const char* fileName, int old, int n, const float* data)
{
  return guard([&] {
    if (!fileName || !data || n < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "wb");
    if (!old) { if (fwrite(data, sizeof(float), (size_t) n, f.fp) != (size_t) n) throw Error(DSR_E_IO, "Could not write vector to %s", fileName); return; }
    fwrite("FVEC", 1, 4, f.fp); f.wbe32(n); { const float z = 0.0f; f.wbefloats(&z, 1); } f.wbefloats(data, (size_t) n);
  });
}

// HTK parameter files (feature.cc:4025-4168): 12-byte header {int32 nSamples, int32 sampPeriod, int16 sampSize, int16 parmKind} and float vectors.
// The reference's flag is named isBigEndian and means "this machine is big endian": when it is false (the default) every field is byte-swapped on
// its way to and from the file, which on today's hosts yields the big-endian files HTK expects.  Restated as written.  parmKind with _C
// (compressed, 02000) or _K (CRC, 010000) is refused (DSR_E_IO), as WriteHTKFeatureFile and HTKFeature::read do.
static inline void swap4(void* p) { unsigned char* b = (unsigned char*) p; std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
static inline void swap2(void* p) { unsigned char* b = (unsigned char*) p; std::swap(b[0], b[1]); }
dsr_status dsr_htk_write(const char* fileName, int nSamples, int sampPeriod, int sampSize, int parmKind, int isBigEndian, const float* data)
{
  return guard([&] {
    if (!fileName || (!data && nSamples > 0) || sampSize < 0 || sampSize % 4) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "wb");
    int32_t ns = nSamples, sp = sampPeriod; int16_t ss = (int16_t) sampSize, pk = (int16_t) parmKind;
    if (!isBigEndian) { swap4(&ns); swap4(&sp); swap2(&ss); swap2(&pk); }
    fwrite(&ns, 1, 4, f.fp); fwrite(&sp, 1, 4, f.fp); fwrite(&ss, 1, 2, f.fp); fwrite(&pk, 1, 2, f.fp);
    if (parmKind & 02000) throw Error(DSR_E_IO, "The file compression is not supported\n");
    if (parmKind & 010000) throw Error(DSR_E_IO, "The CRC is not supported\n");
    const size_t n = (size_t) nSamples * (size_t) (sampSize / 4);
    for (size_t i = 0; i < n; i++) { float v = data[i]; if (!isBigEndian) swap4(&v); if (fwrite(&v, 4, 1, f.fp) != 1) throw Error(DSR_E_IO, "WriteHTKFeatureFile : WriteFloatBinary() failed\n"); }
  });
}
dsr_status dsr_htk_read_header(const char* fileName, int isBigEndian, int* nSamples, int* sampPeriod, int* sampSize, int* parmKind)
{
  return guard([&] {
    if (!fileName) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "rb");
    int32_t ns = 0, sp = 0; int16_t ss = 0, pk = 0;
    if (fread(&ns, 4, 1, f.fp) != 1 || fread(&sp, 4, 1, f.fp) != 1 || fread(&ss, 2, 1, f.fp) != 1 || fread(&pk, 2, 1, f.fp) != 1) throw Error(DSR_E_IO, "cannot read the HTK header of %s", fileName);
    if (!isBigEndian) { swap4(&ns); swap4(&sp); swap2(&ss); swap2(&pk); }
    if (pk & 02000) throw Error(DSR_E_IO, "The file is compressed\nSAVECOMPRESSED=F\n");
    if (pk & 010000) throw Error(DSR_E_IO, "The CRC is not supported\nset SAVEWITHCRC=F\n");
    if (nSamples) *nSamples = ns; if (sampPeriod) *sampPeriod = sp; if (sampSize) *sampSize = ss; if (parmKind) *parmKind = pk;
  });
}
dsr_status dsr_htk_read(const char* fileName, int isBigEndian, float* data, size_t nFloats)
{
  return guard([&] {
    if (!fileName || !data) throw Error(DSR_E_PARAMETER, "bad argument");
    F f(fileName, "rb");
    if (fseek(f.fp, 12, SEEK_SET) != 0) throw Error(DSR_E_IO, "cannot read %s", fileName);
    if (fread(data, 4, nFloats, f.fp) != nFloats) throw Error(DSR_E_ERROR, "ReadFloatBinary() failed\n");
    if (!isBigEndian) for (size_t i = 0; i < nFloats; i++) swap4(&data[i]);
  });
}

}  // extern "C"
