// csrc/k_gmm_mfma.hip -- placeholder until the MFMA candidate search lands (mode 2 of dsr_gmm_score).
#include "common.h"
namespace dsr {
struct GmmModel;
void gmm_score_mfma(GmmModel&, const float*, long, float*, unsigned char*, hipStream_t)
{ throw Error(DSR_E_ERROR, "dsr_gmm_score mode 2 (MFMA) is not built yet"); }
}
