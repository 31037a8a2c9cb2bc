// csrc/gmm_model.h -- the codebook/distribution set as k_gmm.hip and k_gmm_mfma.hip share it (one definition for both translation units).
#pragma once
#include "common.h"
#include <string>
#include <vector>

namespace dsr {

// what the MFMA scoring path hands from its contraction kernel to its near-tie kernel: owned by the model, ONE PER STREAM (two pipes on two
// HIP streams may score with the same model at the same time; within a stream the two launches are ordered)
struct GmmTieScratch { DevBuf<unsigned long long> list; DevBuf<unsigned> count; DevBuf<unsigned> masks; };

struct GmmModel {
  int K = 0, D = 0, G = 0, maxRef = 0;
  std::vector<int> refN, off;
  std::vector<float> mean, ivar, det, val, scale, pi, count;
  std::vector<std::string> cbNames, dsNames;
  DevBuf<int> d_off;                 // [K+1]
  DevBuf<float> d_mean, d_ivar;      // [G][Dp]  (rows padded to Dp = multiple of 4)
  DevBuf<float> d_cst;               // [G] pi+det
  DevBuf<float> d_val, d_scale;      // [G], [K]
  int Dp = 0;
  // MFMA operand image (built lazily)
  bool mfmaReady = false; int KP = 0, GT = 0;
  DevBuf<float> d_A;                 // [GT][KP/2][64]
  DevBuf<float> d_bn;                // the same operand, four consecutive steps of a lane side by side
  DevBuf<int> d_tileCb;              // codebook ids per 32-Gaussian tile (uniform refN=16 path)
  // bound on the magnitude of the expanded form's terms: sum |terms| <= 2 (ivMax |x|^2 + muIvMax) + cstMax for every Gaussian of the model
  float ivMax = 0.0f, termMax = 0.0f;
  PerStream<GmmTieScratch> tie;
};

// k_gmm_sp.hip: the software-pipelined MFMA shape for codebooks of four Gaussians
int gmm_sp_frames();
size_t gmm_sp_lds(const GmmModel& m);
bool gmm_sp_has(int S4, int R);
bool gmm_sp_launch(GmmModel& m, int R, const float* x, long N, float* score, unsigned char* argmin, DevBuf<unsigned>& masks, unsigned long long* tieList, unsigned* tieCount, unsigned cap, hipStream_t st);

}  // namespace dsr
