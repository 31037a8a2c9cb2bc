/*
 * oracle/orc.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the BTK->ASR front-end-to-decode hot path of
 * mmdagent/distantspeechrecognition-mirror.  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this library; the product (libdsr_hip.so)
 * never links or loads it.
 *
 * Pinning status ("parity unpinned" caveat): the reference ships no golden
 * vectors and cannot be built here (needs GSL/sndfile/SWIG and the autoconf
 * generated btk.h/config.h).  What IS pinned:
 *   - complex-float SVD / pseudo-inverse against the reference's in-tree
 *     LINPACK csvdc, compiled from the reference sources into oracle/_ref/
 *   - filterbank conventions through the perfect-reconstruction property of the
 *     reference's shipped Nyquist(M) prototypes on the reference's Headset1.wav
 *   - file formats through byte-level round trips
 * Everything else is restated from the source text and is "parity unpinned".
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- Zelinski post-filter (btk/postfilter/postfilter.cc:8-221,428-493) ---------------- */
int  orc_zelinski_postfilter(const double* X, const double* Y, const double* wq, int C, int T, int F, double alpha, int type, int minFrames,
                             double* out, double* wp1);

void orc_pf_diffuse_noise_model(const double* micPos, int C, int M, double sampleRate, double sspeed, double* R);
int  orc_mccowan_postfilter(const double* X, const double* Y, const double* wq, const double* R, int C, int T, int F, double alpha, int type,
                            int minFrames, double threshold, double* out, double* wp1);

/* ---------------- single-channel WPE dereverberation (btk/dereverberation/dereverberation.cc:28-300) ---------------- */
int  orc_wpe_single(const double* Y, int N, int M, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                    double* out, double* gnOut);

/* ---------------- LPC / MVDR spectral envelopes (btk/feature/lpc.cc, lpc.h) ---------------- */
int  orc_lpc_npoints(int dim);
void orc_lpc_fft_power(float* power, int dim);
void orc_lpc_warp_autocorr(const float* X, int dim, int order, float warp, float* LP, float* E);
void orc_lpc_burg_autocorr(const float* X, int dim, int order, float* A, float* E);
/* method 0 Warp / 1 Burg; kind 0 MVDR envelope / 1 LPC envelope; frames [T][dim] -> out [T][dim/2+1] */
int  orc_lpc_feature(const float* frames, long T, int dim, int order, float warp, int method, int kind, double* out);

/* ---------------- filter banks (btk/modulated/modulated.cc) ---------------- */
/* delayCompensationType: 0 default, 1, 2 (modulated.cc:279-296) */
int  orc_fb_processing_delay(int m, int r, int dctype, int synthesis);
int  orc_fb_lookahead(int m, int r, int dctype, int synthesis);
/* number of analysis frames for nsamp input samples (padZeros source) */
int  orc_analysis_num_frames(int nsamp, int M, int m, int r, int dctype);
/* X: [T][M] interleaved complex double */
void orc_analysis_bank(const float* x, int nsamp, const double* h, int M, int m, int r,
                       int dctype, int gain, double* X);
/* Y: [T][M] complex double in; out: [(T-pd)][D] float; returns number of output blocks */
int  orc_synthesis_bank(const double* Y, int T, const double* g, int M, int m, int r,
                        int dctype, int gain, float* out);
void orc_get_window(int winType, int winLen, double* win);
/* PerfectReconstructionFFTAnalysisBank / ...SynthesisBank (modulated.cc:686-970): 2M bands, prototype length 2M*m */
int  orc_pr_analysis_num_frames(int nsamp, int M, int m, int r);
void orc_pr_analysis_bank(const float* x, int nsamp, const double* h, int M, int m, int r, double* X);
int  orc_pr_synthesis_bank(const double* Y, int T, const double* g, int M, int m, int r, float* out);
/* NormalFFTAnalysisBank (modulated.cc:121-257): X [T][M] */
int  orc_normal_fft_num_frames(int nsamp, int M, int r);
void orc_normal_fft_bank(const float* x, int nsamp, int M, int r, int winType, double* X);

/* ---------------- beamformer (btk/beamformer/beamformer.cc) ---------------- */
void orc_calc_mainlobe(double fs, const double* delays, int C, int M, double* wq /*[M][C][2]*/);
void orc_calc_delays_polar2(float azimuth, float elevation, const double* micpos, int C,
                            double* delays);
void orc_diffuse_noise_model(const double* micpos /*[C][3]*/, int C, int M, double fs,
                             double sspeed, double* R /*[M/2+1][C][C][2]*/);
void orc_divide_nondiag(double* R, int C, int M, float mu);
void orc_diagonal_loading(double* R, int C, int M, float w);
/* complex<float> SVD based pseudo inverse (beamformer.cc:253-305). returns 1 ok / 0 failed */
int  orc_pseudoinverse(const double* A, int n, double* invA, float thr);
int  orc_pseudoinverse_mn(const double* A, int M, int N, double* invA /*[N][M]*/, float thr);

/* SubbandMMI (beamformer.cc:1753-2319): one weight set per source, GSC output of the target, Zelinski post-filter, binary mask */
typedef struct orc_mmi orc_mmi;
orc_mmi* orc_mmi_create(int fftLen, int chanN, int halfBandShift, int targetSourceX, int nSource, int pfType, double alpha);
void orc_mmi_free(orc_mmi*);
void orc_mmi_reset(orc_mmi*);
void orc_mmi_use_binary_mask(orc_mmi*, double avgFactor, unsigned fwidth, unsigned type);
int  orc_mmi_calc_weights(orc_mmi*, double sampleRate, const double* delays /*[nSource][C]*/);
int  orc_mmi_calc_weights_n(orc_mmi*, double sampleRate, const double* delays, unsigned NC);
int  orc_mmi_set_active_weights_f(orc_mmi*, unsigned fbinX, const double* packed /*[nSource][2(C-NC)]*/, int option);
int  orc_mmi_set_hi_active_weights_f(orc_mmi*, unsigned fbinX, const double* pkdWa, const double* pkdwb, int option);
void orc_mmi_get(const orc_mmi*, int srcX, int kind /*0 wq, 1 wl, 2 B, 3 ta, 4 wa*/, double* out);
int  orc_mmi_next(orc_mmi*, const double* X /*[C][Fin] complex*/, int Fin, double* out /*[fftLen] complex*/);
/* LINPACK csvdc, job 11 (orc_svd.c): interleaved complex<float>, column major; s, e hold 2 (n + p) + 2 entries */
int  orc_csvdc(float* x, int ldx, int n, int p, float* s, float* e, float* u, int ldu, float* v, int ldv);
void orc_mvdr_weights(const double* wq, const double* R, int C, int M, double thr,
                      double* w /*[M/2+1][C][2]*/);
/* X: [C][T][M] complex double; W: [M/2+1][C] (DS uses wq[0..M/2]); Y: [T][M] */
void orc_beamform_apply(const double* X, const double* W, int C, int T, int M, double* Y);
/* blocking matrix + GSC (beamformer.cc:398-479, 1251-1287) */
int  orc_blocking_matrix(const double* d /*[C][2]*/, int C, double* B /*[C][C-1][2]*/);
void orc_gsc_apply(const double* X, const double* wq, const double* B, const double* wa,
                   int C, int T, int M, int normalize, double* Y);
/* SubbandGSCRLS (beamformer.cc:1497-1698) */
void orc_gsc_rls(const double* X, const double* wq, const double* B, const double* P0, const double* diagW, int C, int T, int M, double myu,
                 double alpha, int qctype, int adapt, int normalize, double* Y, double* waOut);

/* ---------------- MFCC chain (btk/feature/feature.cc) ---------------- */
int  orc_sample_num_blocks(int nsamp, int blockLen, int shiftLen, int padZeros);
void orc_sample_blocks(const float* x, int nsamp, int blockLen, int shiftLen, int padZeros,
                       float* out /*[T][blockLen]*/);
int  orc_blockconv_num_frames(int nIn /*source blocks*/, int inLen, int blockLen, int shiftLen);
void orc_blockconv(const float* in /*[nIn][inLen]*/, int nIn, int inLen, int blockLen,
                   int shiftLen, float* out);
void orc_preemphasis(const float* in, int T, int L, double mu, float* out);
void orc_hamming(const float* in, int T, int L, float* out);
void orc_fft_feature(const float* in, int T, int L, int fftLen, double* out /*[T][fftLen][2]*/);
void orc_spectral_power(const double* fft, int T, int fftLen, int powN, double* out);
void orc_vtln(const double* pow_, int T, int N, double ratio, double edge, int version,
              double* out);
typedef struct {
  int filterN; int n; /* required input length (_n) */
  int* offset; int* coefN; float** data;
} orc_melbank;
orc_melbank* orc_melbank_create(int powN, float rate, float low, float up, int filterN, int version);
void orc_melbank_free(orc_melbank*);
void orc_mel(const orc_melbank* mb, const double* pow_, int T, int powN, int version, double* out);
void orc_log(const double* mel, int T, int N, double m, double a, int sphinxFlooring, float* out);
void orc_cosine_matrix(int ncep, int nmel, int type, float* C /*[ncep][nmel]*/);
void orc_sgemv_rows(const float* A, int rows, int cols, const float* X, int T, float* Y);
void orc_cmn_batch(const float* in, int T, int N, double devNormFactor, float* out,
                   float* mean, float* var);
void orc_cmn_runon(const float* in, int T, int N, double devNormFactor, float* out);
/* with the per-frame weights of MeanSubtractionFeature(src, weight, ...) (feature.cc:2577-2707); weights NULL = all 1 */
void orc_cmn_batch_w(const float* in, int T, int N, double devNormFactor, const float* weights, float* out, float* mean, float* var);
void orc_cmn_runon_w(const float* in, int T, int N, double devNormFactor, const float* weights, float* out);
int  orc_adjacent(const float* in, int T, int N, int delta, float* out);
/* full chain (config 1): returns number of frames, out [T][outDim] */
typedef struct {
  int blockLen, shiftLen, padZeros; double mu; int fftLen, powN;
  double vtlnRatio, vtlnEdge; int vtlnVersion;
  float rate, low, up; int filterN, melVersion;
  double logM, logA; int ncep, dctType; double devNormFactor; int delta;
  int outDim; const float* lda; /* [outDim][(2delta+1)*ncep] or NULL */
  int sphinxFlooring;           /* LogFeature: floor the mel energies at 1e-5 instead of adding a (feature.cc:2411-2418) */
} orc_mfcc_cfg;
void orc_mfcc_default_cfg(orc_mfcc_cfg* c);
int  orc_mfcc_num_frames(const orc_mfcc_cfg* c, int nsamp);
/* if blocks!=NULL it is used as source frames ([T][blockLen]); stage selects what is returned:
   0=final, 1=cepstra before CMN, 2=after CMN, 3=log-mel, 4=power */
int  orc_mfcc_chain(const orc_mfcc_cfg* c, const float* x, int nsamp, int stage, float* out);
int  orc_mfcc_from_blocks(const orc_mfcc_cfg* c, const float* blocks, int T, int stage, float* out);

/* ---------------- GMM (asr/gaussian/codebookBasic.cc) ---------------- */
typedef struct {
  int K;            /* codebooks */
  int dimN;
  int* refN;        /* [K] */
  int* off;         /* [K+1] gaussian offsets */
  float* mean;      /* [G][dimN] */
  float* ivar;      /* [G][dimN] */
  float* det;       /* [G] */
  float* count;     /* [G] */
  float* pi;        /* [K] */
  float* scale;     /* [K] */
} orc_cbset;
/* score_opt: out score [T][K], argmin [T][K]; val: [G] (-log w) distribution 1:1 codebook */
void orc_gmm_score_opt(const orc_cbset* cb, const float* val, const float* x, int T,
                       float* score, int32_t* argmin);
void orc_gmm_score_all(const orc_cbset* cb, const float* val, const float* x, int T, float* score);
void orc_gmm_log_lhood(const orc_cbset* cb, const float* val, const float* x, int T, float* score, int32_t* argmin);

/* ---------------- WFST + decoder (asr/decoder) ---------------- */
typedef struct orc_wfst orc_wfst;
orc_wfst* orc_wfst_new(void);
void orc_wfst_free(orc_wfst*);
/* returns 0 ok; mirrors _readText/_readBinary numeric-symbol form */
int  orc_wfst_add_arc(orc_wfst*, unsigned s1, unsigned s2, unsigned in, unsigned out, float cost);
int  orc_wfst_add_final(orc_wfst*, unsigned s, float cost);
int  orc_wfst_read(orc_wfst*, const char* file, int binary);
int  orc_wfst_read_dynamic(orc_wfst* g, const char* file, int noSelfLoops);   /* WFSTransducer::read, asr/fsm/fsm.cc:901-986 */
int  orc_wfst_write(const orc_wfst*, const char* file, int binary);
int  orc_wfst_num_nodes(const orc_wfst*);
int  orc_wfst_num_arcs(const orc_wfst*);
/* export in iteration order: node table + CSR arcs (arc order = reference iteration order) */
void orc_wfst_export(const orc_wfst*, unsigned* nodeState, int* nodeFinal, float* nodeCost,
                     int* arcOff, int* arcDst, unsigned* arcIn, unsigned* arcOut, float* arcCost);

typedef struct {
  double beam, lmScale, lmPenalty, silPenalty; unsigned silenceX;
  int dumpTokens;   /* record every frame's token list (list order) in the result */
  int topN;         /* > 0: expand the topN best tokens in score order, no beam (decoder.h:571-581) */
} orc_dec_cfg;
typedef struct {
  double score; float ac, lm; int frames; int reachedFinal; int nArcs;
  int* arcs;       /* best path arc ids (export numbering), first..last, incl. eps arcs */
  int* arcFrames;  /* token frame for each */
  int nWords; unsigned* words; /* output symbols !=0 in order */
  long activeHypos;
  int nActive; int* activeCount; /* per frame |next| after processing */
  double* topScore; /* per frame */
  /* optional per-frame dump of _next in list order (front first) */
  long* dumpOff; int* dumpNode; float* dumpAc; float* dumpLm; int* dumpArc; long dumpN, dumpCap;
  int finalStatesN;  /* decoder.h:598-608 */
} orc_dec_result;
/* _Decoder::lattice() (decoder.h:805-953): nodes are numbered in creation order (0 = the initial node), nodeFinal[i] 1 final / 0 not / -1 unused
   index; edges in creation order with per-node lists in the reference's iteration order (last added first) */
typedef struct {
  int nNodes, capNodes; int* nodeFinal; int* nodeFirstEdge;
  int nEdges, capEdges; int* from; int* to; unsigned* in; unsigned* out; int* start; int* end; double* ac; double* lm; int* nextEdge;
} orc_lattice;
/* label runs of the best path as _Decoder::writeGMM collects them (decoder.h:1018-1102), last run first; n = -1: no best token */
typedef struct { int n; unsigned* inX; int* startX; int* endX; double* score; } orc_gmm_rows;
int  orc_decode_ex(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX, orc_gmm_rows* gmm);
void orc_gmm_rows_free(orc_gmm_rows* R);
int  orc_decode_lat(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX);
int  orc_lattice_write(const orc_lattice* L, const char* file, int writeData);   /* Lattice::write(file, false, writeData) (lattice.cc:715-757) */
void orc_lattice_free(orc_lattice* L);
/* scores: [T][nDist] (cost of distribution d at frame t) */
int  orc_decode(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist,
                orc_dec_result* res);
void orc_dec_result_free(orc_dec_result*);

/* ---------------- big-endian machine independent I/O (btk/common/mach_ind_io.cc) ---------- */
int  orc_cbset_save(const orc_cbset*, const char** names, const char* file);
orc_cbset* orc_cbset_load(const char* file, char*** names);
void orc_cbset_free(orc_cbset*);
int  orc_distset_save(int n, const char** names, const char** cbnames, const int* refN,
                      const float* count, const float* const* val, const char* file);
int  orc_distset_load(const char* file, int* n, char*** names, char*** cbnames, int** refN,
                      float** count, float*** val);

#ifdef __cplusplus
}
#endif
#endif
