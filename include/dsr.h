/*
 * include/dsr.h -- C-ABI of libdsr_hip.so: the MI355X (gfx950) implementation of the
 * BTK -> ASR front-end-to-decode hot path of mmdagent/distantspeechrecognition-mirror.
 *
 * Boundary rules
 *   - extern "C", opaque handles, plain pointers and sizes; no C++/torch types.
 *   - "dev" pointers are device (HBM) addresses owned by the caller; "host" pointers are
 *     ordinary host memory.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - every function returns a dsr_status: 0 = OK, otherwise 1 + the reference's
 *     error_type (btk/common/jexception.h:41-57), so JITERATOR ("end of stream",
 *     which the reference signals by exception) is DSR_E_ITERATOR.  The text of the
 *     last error on the calling thread is returned by dsr_last_error().
 *   - the library has no CPU fallback: without a usable HIP device every compute
 *     entry point fails with DSR_E_INITIALIZATION.
 *
 * Each entry point names the reference interface it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the SWIG/ctypes stubs a maintainer would add.
 */
#ifndef DSR_H
#define DSR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int dsr_status;
enum {
  DSR_OK = 0,
  DSR_E_ERROR = 1,          /* JERROR          */
  DSR_E_ALLOCATION = 2,     /* JALLOCATION     */
  DSR_E_ARITHMETIC = 3,     /* JARITHMETIC     */
  DSR_E_CONSISTENCY = 4,    /* JCONSISTENCY    */
  DSR_E_DIMENSION = 5,      /* JDIMENSION      */
  DSR_E_INDEX = 6,          /* JINDEX          */
  DSR_E_INITIALIZATION = 7, /* JINITIALIZATION */
  DSR_E_IO = 8,             /* JIO             */
  DSR_E_ITERATOR = 9,       /* JITERATOR: end of stream */
  DSR_E_PYTHON = 10,        /* JPYTHON         */
  DSR_E_KEY = 11,           /* JKEY            */
  DSR_E_NUMERIC = 12,       /* JNUMERIC        */
  DSR_E_PARAMETER = 13,     /* JPARAMETER      */
  DSR_E_PARSE = 14,         /* JPARSE          */
  DSR_E_TYPE = 15           /* JTYPE           */
};

const char* dsr_last_error(void);
const char* dsr_version(void);
/* number of HIP devices visible; selects `device` for the calling thread */
dsr_status dsr_device_count(int* n);
dsr_status dsr_set_device(int device);
dsr_status dsr_stream_synchronize(void* stream);
/* utility for callers without their own HIP binding: synchronous device -> host copy */
dsr_status dsr_memcpy_dtoh(void* dst_host, const void* src_dev, size_t bytes, void* stream);

/* =====================================================================================
 * 1. Modulated (uniform DFT) filter banks
 *    replaces OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank
 *    (btk/modulated/modulated.h:291-321, modulated.cc:360-516, 521-674)
 * ===================================================================================== */
typedef struct dsr_fb dsr_fb;
/* prototype: m*M taps (host, double, copied as the reference copies it, modulated.cc:275-276).
   delayCompensationType 0/1/2 (modulated.cc:279-296), gainFactor as modulated.cc:444-448,660-661.
   M must be a power of two in [16, 2048] and D = M >> r >= 1. */
dsr_status dsr_fb_create(const double* prototype, int M, int m, int r, int synthesis,
                         int delayCompensationType, int gainFactor, dsr_fb** out);
void       dsr_fb_destroy(dsr_fb*);
/* frame bookkeeping of the reference (modulated.cc:461-516, 626-664) */
int dsr_fb_analysis_frames(const dsr_fb*, int nsamp);       /* ceil(n/D)-laN+processingDelay */
int dsr_fb_synthesis_blocks(const dsr_fb*, int nframes);    /* nframes-processingDelay (>=0) */
int dsr_fb_processing_delay(const dsr_fb*);
int dsr_fb_block_len(const dsr_fb*);                        /* D = M >> r */
/* Batched analysis over U utterances x C channels.
 *   x_dev     [U][C][sampStride] fp32 samples (SampleFeature with blockLen=shiftLen=D, padZeros)
 *   nsamp_dev [U] int32 valid samples per utterance
 *   X_dev     [U][C][Tmax][M/2+1] complex64 (re,im); frames t >= T_u are zero filled.
 * Only bins 0..M/2 are stored: the input is real, bin M-f is conj(bin f) (beamformer.cc:2609-2631
 * consumes exactly these). */
dsr_status dsr_fb_analysis(const dsr_fb*, const float* x_dev, const int32_t* nsamp_dev, int U, int C,
                           int64_t sampStride, int Tmax, float* X_dev, void* stream);
/* Batched synthesis.
 *   Y_dev      [U][Tmax][M/2+1] complex64, nframes_dev [U] valid frames
 *   y_dev      [U][outStride] fp32: blocks 0..nframes-pd-1 of D samples; the rest zero filled */
dsr_status dsr_fb_synthesis(const dsr_fb*, const float* Y_dev, const int32_t* nframes_dev, int U,
                            int Tmax, int64_t outStride, float* y_dev, void* stream);
/* Analysis bank and fixed-weight subband beamformer in one pass: Y_dev [U][Tmax][M/2+1] = dsr_bf_apply(dsr_fb_analysis(x)) without the channel
 * snapshots X ever being written (OverSampledDFTAnalysisBank::next per channel, modulated.cc:461-516, feeding SubbandDS / SubbandMVDR / SubbandGSC::next,
 * beamformer.cc:1137-1200,1297-1363,2583-2635, whose output is sum_c conj(w[f][c]) X_c[f] with weights that do not change from frame to frame).
 * supported(): M = 256, r = 1, m in {2, 4}, at most 16 channels, no halfBandShift, not the adapting SubbandGSCRLS; otherwise call the two steps. */
struct dsr_bf;
int        dsr_fb_analysis_beamform_supported(const dsr_fb*, const struct dsr_bf*);
dsr_status dsr_fb_analysis_beamform(const dsr_fb*, struct dsr_bf*, const float* x_dev, const int32_t* nsamp_dev, int U, int C, int64_t sampStride,
                                    int Tmax, float* Y_dev, void* stream);

/* Block-wise processing of long streams (BASELINE configs[4]: 10-minute streams handed over in 10-second blocks).  The reference operators are
 * streaming by construction: they keep ring buffers of the last m*M samples (analysis: _RealBuffer, modulated.h:79-163, modulated.cc:400-452)
 * and of the last R*m subband frames (synthesis: modulated.cc:586-664).  A dsr_fb_state carries exactly that from one call to the next
 * (m*M - D samples per (stream, channel); R*m - 1 subband frames per stream), so that the frames of all blocks together are the frames of the whole
 * stream.  All U streams advance in step: every block but a stream's last holds a multiple of D samples.
 *   analysis_block : x_dev [U][C][sampStride] the block's new samples, nsamp_dev [U]; last != 0 appends the processingDelay zero-input frames
 *                    (modulated.cc:493-501).  Frames written per stream: dsr_fb_analysis_block_frames(plan, state, nsamp, last), asked BEFORE the call
 *                    (the stream's first block spends the look-ahead of delayCompensationType 2).  X_dev [U][C][Tmax][M/2+1].
 *   synthesis_block: Y_dev [U][Tmax][M/2+1], nframes_dev [U] (nframesHostMax = their maximum); output blocks per stream:
 *                    dsr_fb_synthesis_block_blocks(plan, state, nframes), asked before the call (the first call keeps processingDelay frames of
 *                    look-ahead back, modulated.cc:631-634, and needs at least processingDelay + R*m frames).  y_dev [U][outStride]. */
typedef struct dsr_fb_state dsr_fb_state;
dsr_status dsr_fb_state_create(const dsr_fb*, int U, int C /* ignored for a synthesis plan */, dsr_fb_state** out);
void       dsr_fb_state_destroy(dsr_fb_state*);
dsr_status dsr_fb_state_reset(dsr_fb_state*);            /* the next block starts new streams */
int        dsr_fb_analysis_block_frames(const dsr_fb*, const dsr_fb_state*, int nsampBlock, int last);
dsr_status dsr_fb_analysis_block(const dsr_fb*, dsr_fb_state*, const float* x_dev, const int32_t* nsamp_dev, int U, int C, int64_t sampStride,
                                 int last, int Tmax, float* X_dev, void* stream);
int        dsr_fb_synthesis_block_blocks(const dsr_fb*, const dsr_fb_state*, int nframesBlock);
dsr_status dsr_fb_synthesis_block(const dsr_fb*, dsr_fb_state*, const float* Y_dev, const int32_t* nframes_dev, int nframesHostMax, int U, int Tmax,
                                  int64_t outStride, float* y_dev, void* stream);

/* =====================================================================================
 * 2. Subband beamformers
 *    replaces beamformerWeights / SubbandDS / SubbandGSC / SubbandMVDR
 *    (btk/beamformer/beamformer.h:49-118,120-223,316-383; beamformer.cc:531-594,1137-1200,
 *     1297-1447,2321-2635)
 * ===================================================================================== */
typedef struct dsr_bf dsr_bf;
dsr_status dsr_bf_create(int fftLen, int chanN, int halfBandShift, dsr_bf** out);
void       dsr_bf_destroy(dsr_bf*);
int        dsr_bf_fft_len(const dsr_bf*);
int        dsr_bf_chan_n(const dsr_bf*);
/* halfBandShift == true (beamformer.cc:544-555,1159-1175,1321-1330): all fftLen bins are computed independently -- no conjugate mirror, no special
   bin 0; the snapshot and output arrays of dsr_bf_apply then carry dsr_bf_bins() = fftLen bins per frame instead of fftLen/2+1.  Delay-and-sum and
   GSC only: SubbandMVDR refuses the flag (:2324-2327), SubbandGSCRLS::next says "not yet implemented" (:1580-1583) -- both kept. */
int        dsr_bf_half_band_shift(const dsr_bf*);
int        dsr_bf_is_adaptive(const dsr_bf*);      /* 1 after dsr_bf_rls_config: the output depends on the frames before (SubbandGSCRLS) */
int        dsr_bf_bins(const dsr_bf*);
/* calcArrayManifoldVectors (beamformer.cc:531-594): delays[chanN] seconds */
dsr_status dsr_bf_calc_array_manifold(dsr_bf*, double sampleRate, const double* delays);
/* calcDelaysPolar2 of the reference driver (btk/src/superdirectiveBeamformer.cc:118-137) */
dsr_status dsr_calc_delays_polar2(float azimuth, float elevation, const double* micPos /*[C][3]*/,
                                  int chanN, double* delays);
/* SubbandMVDR::setDiffuseNoiseModel / divideAllNonDiagonalElements / setAllLevelsOfDiagonalLoading /
   setNoiseSpatialSpectralMatrix / calcMVDRWeights (beamformer.cc:2392-2581, beamformer.h:362-378) */
dsr_status dsr_bf_set_diffuse_noise_model(dsr_bf*, const double* micPos /*[C][3]*/, double sampleRate, double sspeed);
dsr_status dsr_bf_divide_nondiagonal(dsr_bf*, float myu);
dsr_status dsr_bf_diagonal_loading(dsr_bf*, float diagonalWeight);
dsr_status dsr_bf_set_noise_matrix(dsr_bf*, int fbinX, const double* Rnn /*[C][C] complex*/);
dsr_status dsr_bf_calc_mvdr_weights(dsr_bf*, double sampleRate, double dThreshold);
/* pseudoinverse(A, invA, dThreshold) (beamformer.cc:253-305): LINPACK csvdc (btk/matrix/linpack_c.cc:9518, job 11) in complex<float>,
   V diag(1/s) U^H with singular values below dThreshold dropped.  A [rows][cols] complex128 row major (host) -> invA [cols][rows];
   *ok = the reference's return value (0: a singular value was dropped or csvdc did not converge); svals (optional) min(rows, cols) floats. */
dsr_status dsr_pseudoinverse(const double* A, int rows, int cols, float dThreshold, double* invA, int* ok, float* svals);
/* SubbandGSC: calcGSCWeights (blocking matrices), setActiveWeights_f, zeroActiveWeights
   (beamformer.cc:1373-1447, 761-799, 398-479) */
dsr_status dsr_bf_calc_gsc_weights(dsr_bf*, double sampleRate, const double* delays);
dsr_status dsr_bf_set_active_weights(dsr_bf*, int fbinX, const double* packedWeight /*2*(C-1)*/);
dsr_status dsr_bf_zero_active_weights(dsr_bf*);
/* SubbandMVDRGSC (beamformer.h:394-425, beamformer.cc:2637-2817): mode 4 of dsr_bf_select = w_mvdr - wl with wl = B wa as cached by
 * the last setActiveWeights_f / zeroActiveWeights.  calcBlockingMatrix1(sampleRate, delays) is dsr_bf_calc_gsc_weights; calcBlockingMatrix2(),
 * upgradeBlockingMatrix(), blockingMatrixOutput(outChanX) for a batch (Y_dev [U][Tmax][M/2+1]) */
dsr_status dsr_bf_calc_blocking_matrix2(dsr_bf*);
dsr_status dsr_bf_upgrade_blocking_matrix(dsr_bf*);
dsr_status dsr_bf_blocking_matrix_output(dsr_bf*, const float* X_dev, int U, int Tmax, int outChanX, float* Y_dev, void* stream);
/* SubbandGSCRLS(fftLen, halfBandShift, myu, sigma2) (beamformer.h:213-262, beamformer.cc:1497-1698; SURVEY 8f rank 3, first operator): after
 * rls_config the object's apply is the GSC whose active weights follow a recursive-least-squares update after every frame; every utterance
 * of a batch starts from the precision matrices set here and zero active weights (the reference keeps adapting across reset()).
 * initPrecisionMatrix / setPrecisionMatrix / setQuadraticConstraint(alpha, qctype: 0 none, 1 constant norm, 2 threshold) /
 * updateActiveWeightVecotrs(flag) */
dsr_status dsr_bf_rls_config(dsr_bf*, float myu, float sigma2);
dsr_status dsr_bf_rls_init_precision(dsr_bf*, float sigma2);
dsr_status dsr_bf_rls_set_precision(dsr_bf*, int fbinX, const double* Pz /* [C-1][C-1] complex128 */);
dsr_status dsr_bf_rls_quadratic_constraint(dsr_bf*, float alpha, int qctype);
dsr_status dsr_bf_rls_adapt(dsr_bf*, int flag);
/* X_dev [U][C][Tmax][M/2+1] complex64 -> Y_dev [U][Tmax][M/2+1]; wa_out_dev (optional) [U][M/2+1][C-1] complex128: the final active weights.
 * nframes_dev (optional) [U]: utterance u is adapted over its first nframes[u] frames only -- the reference stops adapting at the stream's last
 * frame (beamformer.cc:1552-1612) -- and the rest of its rows is zero.  chanN <= 64. */
dsr_status dsr_bf_gsc_rls(dsr_bf*, const float* X_dev, const int32_t* nframes_dev, int U, int Tmax, float* Y_dev, double* wa_out_dev, void* stream);
/* Carried adaptation state.  The reference object keeps its precision matrices and active weights across reset() (beamformer.cc:1552-1700: only
 * initPrecisionMatrix / setPrecisionMatrix re-seed them).  carry = 1: every dsr_bf_gsc_rls / dsr_bf_apply(_frames) call of the same U continues from
 * the state the previous one left -- block-wise processing of long streams, stream u of one call = stream u of the next; rls_reset_state (and the
 * precision-matrix setters) make the next call start from P0 and zero weights again.  carry = 0 (default): every call starts afresh. */
dsr_status dsr_bf_rls_carry(dsr_bf*, int on);
dsr_status dsr_bf_rls_reset_state(dsr_bf*);
/* which weight set `apply` uses: 0 = delay-and-sum wq, 1 = MVDR, 2 = GSC (wq - B wa), 3 = GSC normalised, 4 = MVDR-GSC (w_mvdr - wl) */
dsr_status dsr_bf_select(dsr_bf*, int mode);
/* read back host copies: kind 0 = wq [fftLen][C], 1 = mvdr [fftLen/2+1][C], 2 = R [fftLen/2+1][C][C],
   3 = blocking matrix [fftLen][C][C-1], 4 = effective weights in use [fftLen/2+1][C]; complex double */
dsr_status dsr_bf_get(const dsr_bf*, int kind, double* out, size_t outDoubles);
/* Y[u][t][f] = w_f^H X[u][:][t][f], f = 0..M/2 (SubbandDS::next / SubbandMVDR::next / SubbandGSC::next) */
dsr_status dsr_bf_apply(dsr_bf*, const float* X_dev, int U, int Tmax, float* Y_dev, void* stream);
/* the same with per-utterance frame counts nframes_dev [U]: rows t >= nframes[u] are zero; an adapting (SubbandGSCRLS) object adapts on the valid frames only */
dsr_status dsr_bf_apply_frames(dsr_bf*, const float* X_dev, const int32_t* nframes_dev, int U, int Tmax, float* Y_dev, void* stream);

/* =====================================================================================
 * 3. MFCC feature chain
 *    replaces SampleFeature(block framing) -> PreemphasisFeature -> HammingFeature -> FFTFeature ->
 *    SpectralPowerFeature -> VTLNFeature -> MelFeature -> LogFeature -> CepstralFeature ->
 *    StorageFeature -> MeanSubtractionFeature -> AdjacentFeature -> LinearTransformFeature
 *    (btk/feature/feature.cc:610-659,1154-1355,1705-2298,2398-2503,2530-2987)
 * ===================================================================================== */
typedef struct {
  int blockLen, shiftLen, padZeros;   /* SampleFeature (feature.i:526-528): 320,160,false */
  double mu;                          /* PreemphasisFeature: 0.95; <0 disables the operator */
  int fftLen, powN;                   /* 512, 257 */
  double vtlnRatio, vtlnEdge; int vtlnVersion;   /* 1.0,1.0,1; version 0 disables VTLN */
  float rate, low, up; int filterN, melVersion;  /* 16000,0,0(->rate/2),30,1 */
  double logM, logA; int sphinxFlooring;         /* 1,1,0 */
  int ncep, dctType;                  /* 13, 1 */
  int cmnMode; double devNormFactor;  /* 0 none, 1 batch, 2 run-on (feature.cc:2573-2744) */
  int delta;                          /* AdjacentFeature: 7 (0 disables) */
  int outDim;                         /* LinearTransformFeature rows; 0 disables */
} dsr_mfcc_cfg;
void dsr_mfcc_default_cfg(dsr_mfcc_cfg*);
typedef struct dsr_mfcc dsr_mfcc;
/* lda: [outDim][(2delta+1)*ncep] row major fp32 (host) or NULL when outDim == 0 */
dsr_status dsr_mfcc_create(const dsr_mfcc_cfg*, const float* lda, dsr_mfcc** out);
void       dsr_mfcc_destroy(dsr_mfcc*);
int dsr_mfcc_frames(const dsr_mfcc*, int nsamp);   /* frames the chain yields for nsamp samples */
int dsr_mfcc_out_dim(const dsr_mfcc*);
/* y_dev [U][sampStride] fp32, nsamp_dev [U]; feat_dev [U][Tmax][outDim] fp32 (rows >= T_u zero).
   stage: 0 = final, 1 = cepstra before CMN, 2 = after CMN, 3 = log-mel, 4 = power (as float) */
dsr_status dsr_mfcc_run(dsr_mfcc*, const float* y_dev, const int32_t* nsamp_dev, int U, int64_t sampStride,
                        int Tmax, int stage, float* feat_dev, void* stream);

/* =====================================================================================
 * 4. Diagonal-covariance GMM scoring
 *    replaces CodebookSetBasic / DistribSetBasic and Distrib::score
 *    (asr/gaussian/codebookBasic.cc:258-309,431-554,645-766,906-960; distribBasic.cc:32-41,103-166)
 * ===================================================================================== */
typedef struct dsr_gmm dsr_gmm;
/* K codebooks; refN[k] Gaussians (<= 256, codebookBasic.h:41); mean/ivar [G][dimN]; det [G];
   val [G] = -log w of the (1:1) distribution; scale[K] or NULL (=1).  All host pointers. */
dsr_status dsr_gmm_create(int K, int dimN, const int32_t* refN, const float* mean, const float* ivar,
                          const float* det, const float* val, const float* scale, dsr_gmm** out);
/* big-endian model files written by CodebookSetBasic::save / DistribSetBasic::save */
dsr_status dsr_gmm_load(const char* codebookFile, const char* distribFile, dsr_gmm** out);
dsr_status dsr_gmm_save(const dsr_gmm*, const char* codebookFile, const char* distribFile);
/* the older (Janus) codebook-set format: dsr_gmm_load reads it when the file does not start with CodebookMagic (CodebookSetBasic::load
   :934-957 -> CodebookBasic::loadOld :311-350; a uniform covariance type or, with -1, a count and a type per Gaussian; "Wrong covariance type."
   for anything but diagonal; the compressed mode is refused as in the reference).  save_janus = save(filename, janusFormat = true) (:352-383,962-983). */
dsr_status dsr_gmm_save_janus(const dsr_gmm*, const char* codebookFile, const char* distribFile /* may be NULL */);
void       dsr_gmm_destroy(dsr_gmm*);
int dsr_gmm_num_dists(const dsr_gmm*);
int dsr_gmm_dim(const dsr_gmm*);
/* names of the set (file order) and DistribSet::find(name) / index(key) (asr/gaussian/distribBasic.h:183-190): DSR_E_KEY when absent */
const char* dsr_gmm_dist_name(const dsr_gmm*, int distX);
const char* dsr_gmm_codebook_name(const dsr_gmm*, int cbX);
dsr_status dsr_gmm_find_dist(const dsr_gmm*, const char* name, int* distX);
/* x_dev [N][dimN] fp32 -> score_dev [N][K] fp32 (cost), argmin_dev [N][K] u8 or NULL.
   mode 0: _scoreOpt nearest Gaussian, bit-exact reference order; mode 1: _scoreAll log-sum;
   mode 2: _scoreOpt through the fp32-MFMA contraction of the expanded quadratic: argmin equals mode 0's on every frame (every codebook whose two best
   lie inside the expanded form's rounding bound is re-scored in reference order); cost within rel 2e-6 of mode 0 on well-conditioned models
   (re-scored entries: mode 0's bits), never worse than 1e-3 */
dsr_status dsr_gmm_score(dsr_gmm*, const float* x_dev, int64_t N, int mode, float* score_dev,
                         uint8_t* argmin_dev, void* stream);
/* CodebookBasic::logLhood(frame, val) (asr/gaussian/codebookBasic.cc:557-609) for every (frame, codebook): the nearest Gaussian as _scoreOpt finds it,
 * finished as that method does -- 0.5 * min, + val[argmin] when useVal (the reference's val != NULL), no codebook scale.  Bit exact (reference order). */
dsr_status dsr_gmm_log_lhood(dsr_gmm*, const float* x_dev, int64_t N, int useVal, float* score_dev, uint8_t* argmin_dev, void* stream);

/* =====================================================================================
 * 5. Static decoding graph + Viterbi token passing
 *    replaces WFSTFlyWeight (asr/decoder/wfstFlyWeight.h:47-252, .cc:63-139,299-463) and
 *    DecoderFlyWeight / _Decoder (asr/decoder/decoder.h:325-1102,1127-1139; decoder.i:147-199)
 * ===================================================================================== */
/* Lexicon (asr/dictionary/distribTree.h:40-65, distribTree.cc:36-133): symbol <-> index, indices = line order of the file (its index column is
   ignored), ';' comment lines, a repeated symbol is skipped; index() of an unknown symbol is DSR_E_KEY (List::index, btk/common/mlist.h:109-114)
   unless create != 0 */
typedef struct dsr_lexicon dsr_lexicon;
dsr_status dsr_lexicon_create(const char* name, const char* fileName /* "" or NULL: empty */, dsr_lexicon** out);
void       dsr_lexicon_destroy(dsr_lexicon*);
dsr_status dsr_lexicon_read(dsr_lexicon*, const char* fileName);
dsr_status dsr_lexicon_write(const dsr_lexicon*, const char* fileName, int writeHeader);
dsr_status dsr_lexicon_clear(dsr_lexicon*);
int        dsr_lexicon_size(const dsr_lexicon*);
const char* dsr_lexicon_name(const dsr_lexicon*);
int        dsr_lexicon_is_present(const dsr_lexicon*, const char* symbol);
dsr_status dsr_lexicon_index(dsr_lexicon*, const char* symbol, int create, unsigned* index);
dsr_status dsr_lexicon_symbol(const dsr_lexicon*, unsigned index, const char** symbol /* borrowed */);

typedef struct dsr_wfst dsr_wfst;
dsr_status dsr_wfst_create(dsr_wfst** out);
/* WFSTFlyWeight(statelex, inlex, outlex) (decoder.i:52-70): borrowed lexica; the text reader looks fields that are not numbers up in them
   (wfstFlyWeight.cc:311-347); inputLexicon() / outputLexicon() / stateLexicon(); hasFinalState() */
dsr_status dsr_wfst_set_lexicons(dsr_wfst*, dsr_lexicon* stateLex, dsr_lexicon* inputLex, dsr_lexicon* outputLex);
dsr_lexicon* dsr_wfst_state_lexicon(const dsr_wfst*);
dsr_lexicon* dsr_wfst_input_lexicon(const dsr_wfst*);
dsr_lexicon* dsr_wfst_output_lexicon(const dsr_wfst*);
int        dsr_wfst_has_final_state(const dsr_wfst*);
/* WFSTFlyWeightSortedOutput (asr/decoder/wfstFlyWeight.h:403-424, Node::_addEdgeForce wfstFlyWeight.cc:754-776): every node keeps its arcs ordered by
 * (output, input), a new arc in front of the first one that is not smaller.  The container DecoderWordTrace takes.  Call on an empty transducer. */
dsr_status dsr_wfst_set_sorted_output(dsr_wfst*, int on);
void       dsr_wfst_destroy(dsr_wfst*);
dsr_status dsr_wfst_read(dsr_wfst*, const char* fileName, int binary);    /* WFSTFlyWeight::read */
/* the dynamic container's text reader, WFSTransducer::read(fileName, noSelfLoops) (asr/fsm/fsm.cc:901-986): same node/arc
   order as the fly-weight reader; noSelfLoops != 0 skips every self loop (:945).  Graph for Decoder (decoder.h:1107-1125). */
dsr_status dsr_wfst_read_dynamic(dsr_wfst*, const char* fileName, int noSelfLoops);
dsr_status dsr_wfst_write(const dsr_wfst*, const char* fileName, int binary);
/* WFSTFlyWeight::write(fileName, binary, useSymbols) (asr/decoder/wfstFlyWeight.cc:415-463): useSymbols != 0 writes every arc through the lexica set with
 * dsr_wfst_set_lexicons (Edge::write :499-516: "%25s  %25s  %10s  %20s" with a non-empty state lexicon, "%10d  %10d  %10s  %20s" without; a cost below
 * 1e-4 in magnitude is left out); final-state lines -- and, with binary, the end marker -- stay numeric, as the reference writes them.  DSR_E_KEY without lexica. */
dsr_status dsr_wfst_write_symbols(const dsr_wfst*, const char* fileName, int binary, int useSymbols);
/* WFSTFlyWeight::reverse(wfst) (asr/decoder/wfstFlyWeight.cc:141-213): dst becomes src with every arc turned round -- a super-initial node (index
 * _MaximumIndex - 3 = 536870908) with an epsilon arc to each of src's final nodes carrying that node's cost, src's initial state as the only final node.
 * WFSTFlyWeight::reverseRead(fileName) (:215-297): the same from a text file (its first arc's source becomes the final node; a final-state line must come
 * after the arcs that mention the state: DSR_E_KEY "No state %u exists." otherwise, as the reference's find() without create). */
dsr_status dsr_wfst_reverse(dsr_wfst* dst, const dsr_wfst* src);
dsr_status dsr_wfst_reverse_read(dsr_wfst*, const char* fileName);
dsr_status dsr_wfst_add_arc(dsr_wfst*, unsigned s1, unsigned s2, unsigned input, unsigned output, float cost);
dsr_status dsr_wfst_add_final(dsr_wfst*, unsigned state, float cost);
int dsr_wfst_num_nodes(const dsr_wfst*);
int dsr_wfst_num_arcs(const dsr_wfst*);
/* iteration-order export (node 0 = initial; arcs CSR in the order Node::Iterator visits them) */
dsr_status dsr_wfst_export(const dsr_wfst*, uint32_t* nodeState, int32_t* nodeFinal, float* nodeCost,
                           int32_t* arcOff, int32_t* arcDst, uint32_t* arcIn, uint32_t* arcOut, float* arcCost);

typedef struct {
  double beam, lmScale, lmPenalty, silPenalty;   /* decoder.i:191-199: 100, 12, 0, 0 */
  uint32_t silenceX;        /* input-lexicon index of silSymbol (decoder.h:740-745) */
  int maxActive;            /* token capacity per frame  (0 = default 65536)  */
  int maxCandidates;        /* placements per frame      (0 = default 8*maxActive) */
  int64_t arenaTokens;      /* back-pointer records per utterance (0 = default 64 * frames * 1024) */
  int streams;              /* concurrent utterance slots (0 = default 2 per CU) */
  int topN;                 /* decoder.i:198 (default 0).  > 0: _processFrame expands the topN best tokens of the list, in the order of their scores
                               (SortedIterator, decoder.h:298-320; ties in list order, which std::sort leaves open) and applies no beam (:571-581) */
  int64_t latticeTokens;    /* generateLattice (decoder.i:199): > 0 keeps every placement of every frame, at most this many per utterance, for
                               dsr_decoder_lattice(); 0 (default here; the reference always builds its 'worse' chains, decoder.h:1113-1114) = 1-best only */
  /* DecoderWordTrace (asr/decoder/decoder.h:1146-1304, decoder.cc:126-470; decoder.i:201-260), wordTrace != 0: the search over a WFSTFlyWeightSortedOutput
   * whose tokens carry word traces instead of back pointers (float scores compared after rounding, decoder.cc:213-267; the end expansion's float final
   * costs, :185-201).  Results: score / ac / lm / finalStatesN / activeHypos as usual; arcs_out holds the best token's own arc (bestHypo walks prev(), which
   * these tokens do not have: one symbol); words_out / nWords the words along its word traces.  wordTraceLattice = the reference's generateLattice
   * (default 1, as in the reference): that search dereferences a null word trace in the shipped code (:239) and is refused at decode (DSR_E_CONSISTENCY);
   * set it to 0 for the 1-best search.  propagateN / fastHash only steer the refused merge and are kept for the signature; insertSilence: :414-416;
   * wordTraces: word-trace records per utterance (0 = 2^20). */
  int wordTrace, propagateN, fastHash, insertSilence, wordTraceLattice;
  int64_t wordTraces;
} dsr_decoder_cfg;
void dsr_decoder_default_cfg(dsr_decoder_cfg*);
typedef struct dsr_decoder dsr_decoder;
dsr_status dsr_decoder_create(const dsr_decoder_cfg*, dsr_decoder** out);
void       dsr_decoder_destroy(dsr_decoder*);
dsr_status dsr_decoder_set(dsr_decoder*, const dsr_wfst*);               /* DecoderFlyWeight::set */
dsr_status dsr_decoder_set_beam(dsr_decoder*, double beam);
/* _Decoder::setTokenMemoryLimit(limit) (asr/decoder/decoder.h:396) caps the reference's Token memory pool.  Tokens here live in per-slot arrays sized by
 * dsr_decoder_cfg (maxActive, maxCandidates, arenaTokens); an utterance that outgrows them gets DSR_E_ALLOCATION in its result.  There is no pool to
 * limit: the value is accepted and kept (dsr_decoder_token_memory_limit) so that drivers that set it run unchanged. */
dsr_status dsr_decoder_set_token_memory_limit(dsr_decoder*, unsigned limit);
unsigned   dsr_decoder_token_memory_limit(const dsr_decoder*);
/* DecoderFlyWeight::set(wfst) with the symbol look-ups of _Decoder::_set (decoder.h:740-745): silSymbol in the input lexicon (-> cfg.silenceX),
   eosSymbol in the output lexicon; a missing symbol is DSR_E_KEY as in the reference.  NULL symbols are not looked up. */
dsr_status dsr_decoder_set_symbols(dsr_decoder*, const dsr_wfst*, const char* silSymbol, const char* eosSymbol);
uint32_t   dsr_decoder_eos_index(const dsr_decoder*);
/* Results of the last collected decode, utterance u of its batch (the decode must have been collected with paths):
 *   best_hypo  bestHypo(useInputSymbols) (decoder.h:748-773): output symbols != 0 along the best path, or the input symbols != 0 with repetitions
 *              dropped as the reference drops them (walking from the path's end: a symbol is kept when it differs from the one kept after it),
 *              every symbol followed by one blank;
 *   best_path  bestPath() (decoder.h:775-797): the names of the distributions along the path = input symbols != 0, one per line; *count = how many;
 *   path_ids   the same sequences as ids (which 0 outputs, 1 inputs as bestHypo(true), 2 inputs as bestPath) -- no lexicon needed;
 *   final_states_n  finalStatesN() (:598-608);   trace_back_succeeded  traceBackSucceeded() (:611-637).
 * buf may be NULL to ask for the size (*need, including the terminating NUL). */
dsr_status dsr_decoder_best_hypo(const dsr_decoder*, int u, int useInputSymbols, char* buf, size_t cap, size_t* need);
dsr_status dsr_decoder_best_path(const dsr_decoder*, int u, char* buf, size_t cap, size_t* need, int* count);
dsr_status dsr_decoder_path_ids(const dsr_decoder*, int u, int which, uint32_t* ids, int cap, int* n);
dsr_status dsr_decoder_final_states_n(const dsr_decoder*, int u, int* n);
dsr_status dsr_decoder_trace_back_succeeded(const dsr_decoder*, int u, int* ok);
typedef struct {
  double  score;        /* decode() return value: double(ac)+double(lm) of the best token */
  float   ac, lm;
  int32_t frames;       /* _frameX after decode (= T-1) */
  int32_t reachedFinal; /* traceBackSucceeded() */
  int32_t nArcs;        /* arcs on the best path incl. epsilon arcs */
  int32_t nWords;       /* output symbols != 0 */
  int32_t status;       /* per-utterance dsr_status (capacity overflow => DSR_E_ALLOCATION) */
  int32_t maxActiveSeen;
  int64_t activeHypos;  /* sum over frames of |_next| (decoder.h:413) */
  int64_t placements;   /* calls of _placeOnList over the utterance (expanded arcs incl. the end expansion) */
  int64_t registerFrames; /* diagnostics: frames that ran on the decoder kernel's register path (rest: memory path) */
  int32_t finalStatesN; /* finalStatesN() (decoder.h:598-608): tokens of _next in a final state after _expandToEnd */
  int32_t reserved_;
} dsr_decode_result;
/* Batched decode.  score_dev [U][Tmax][nDist] fp32 costs (row t = Distrib::score(t)), nframes_dev [U].
 * Host outputs: res[U]; arcs_out [U][maxPath] (export arc ids, first..last), words_out [U][maxPath]
 * (bestHypo output ids).  arcs_out/words_out may be NULL.
 * Scheduling (results do not depend on it): a batch of more utterances than the device has compute units is decoded in segments of 125 frames, all
 * utterances advancing together, so that they end together (DSR_VITERBI_SEG=<frames> changes the segment, 0 decodes every utterance in one go);
 * a capacity that runs out is per utterance either way (DSR_E_ALLOCATION in its status) -- cfg.arenaTokens then bounds the batch's POOL of
 * back-pointer records, U/8 x (arenaTokens or 8192 x (Tmax+2)) at least, instead of one utterance's. */
dsr_status dsr_decoder_decode_batch(dsr_decoder*, const float* score_dev, const int32_t* nframes_dev, int U,
                                    int Tmax, int nDist, dsr_decode_result* res, int32_t* arcs_out,
                                    uint32_t* words_out, int maxPath, void* stream);
/* The same in two halves: launch enqueues the decode (and the copy of its results to pinned staging memory) on
 * `stream` and returns; collect waits for it and fills the host outputs.  One launch in flight per decoder object. */
dsr_status dsr_decoder_decode_launch(dsr_decoder*, const float* score_dev, const int32_t* nframes_dev, int U,
                                     int Tmax, int nDist, int maxPath, int want_paths, void* stream);
dsr_status dsr_decoder_decode_collect(dsr_decoder*, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out);
/* Lattice generation: _Decoder::lattice(), _majorTrace, _minorTrace, _findLNode (asr/decoder/decoder.h:805-953) over the 'worse' chains of
 * _placeOnList (:531-541), for utterance u of the last decode of a decoder created with cfg.latticeTokens > 0 (DSR_E_CONSISTENCY otherwise: the
 * reference's "Must enable lattice generation during decoding.").  eosX: output-lexicon index of eosSymbol (decoder.h:740-745), used when no token
 * reached a final state (:843-851).  Nodes are numbered as the reference numbers them (0 = the initial node, then creation order); edges come in
 * creation order: from, to, input, output, first and last frame, acoustic score and LM score with the penalties and the LM scale taken out
 * (:921-923).  The lattice object is host memory, independent of the decoder afterwards. */
typedef struct dsr_lattice dsr_lattice;
dsr_status dsr_decoder_lattice(dsr_decoder*, int u, uint32_t eosX, dsr_lattice** out);
/* _Decoder::writeGMM(conv, channel, spk, utt, cfrom, score, fileName, frameInterval) (asr/decoder/decoder.h:1018-1102, decoder.i:177-178): the 1-best
 * path of utterance u as runs of equal input symbols -- "# utt cfrom score", then "conv channel start duration label score" per run, the
 * end-of-sentence label skipped; the score column as the shipped code computes it.  Needs lattice bookkeeping (latticeTokens > 0) and the
 * symbols of dsr_decoder_set_symbols; fileName NULL or "": stdout, otherwise the file is appended to.  (writeCTM is not supported by the
 * reference's decoder template either, decoder.h:399-401.) */
dsr_status dsr_decoder_write_gmm(dsr_decoder*, int u, const char* conv, const char* channel, const char* spk, const char* utt, double cfrom, double score,
                                 const char* fileName, double frameInterval);
void       dsr_lattice_destroy(dsr_lattice*);
int        dsr_lattice_num_nodes(const dsr_lattice*);
int        dsr_lattice_num_edges(const dsr_lattice*);
int        dsr_lattice_final_states_n(const dsr_lattice*);      /* finalStatesN() (decoder.h:598-608) */
dsr_status dsr_lattice_get(const dsr_lattice*, int32_t* nodeFinal, int32_t* from, int32_t* to, uint32_t* in, uint32_t* out, int32_t* start,
                           int32_t* end, double* ac, double* lm);   /* any pointer may be NULL */
/* Lattice::write(fileName, useSymbols = false, writeData) (asr/lattice/lattice.cc:715-757); a cyclic lattice is DSR_E_CONSISTENCY (:862-864).
 * Links print under the nodes' current indices (prune/purge renumber them) with their cost when it is not zero (fsm.cc:1171-1178); the data line
 * is "start end ac lm gamma" (lattice.h:151-154).  fileName "" = stdout. */
dsr_status dsr_lattice_write(dsr_lattice*, const char* fileName, int writeData);
/* The operations of asr/lattice's Lattice (lattice.i:79-123) on a lattice object -- the decoder's, an unpacked one or one read from a file.  Host
 * work, as in the reference (a lattice is hundreds to thousands of links).  The object keeps what the reference's keeps between calls: rescoring
 * tokens, probabilities, posteriors, the cached topological order, and _acScale/_lmScale/penalties of the last call.
 *   read          WFST<..>::read(fileName, noSelfLoops, readData) (asr/fsm/fsm.h:3787-3873): the first state named is the initial node; symbols that
 *                 are not numbers are looked up in the lexica (either may be NULL: DSR_E_KEY then)
 *   rescore       Lattice::rescore(lmScale, lmPenalty, silPenalty, silSymbol) (lattice.cc:122-171): best-token pass in topological order with the
 *                 acoustic scale of the last gammaProbs (1.0 before); silenceX = inputLexicon()->index(silSymbol), resolved by the caller
 *   best_hypo     Lattice::bestHypo(useInputSymbols) (:281-306) as symbol indices, first symbol first (the reference joins them with " ")
 *   gamma_probs   Lattice::gammaProbs(acScale, lmScale, lmPenalty, silPenalty, silSymbol) (:309-379): forward/backward over the sorted nodes,
 *                 posterior (as a negative log) per link; returns the lattice's forward probability; the reference's consistency errors
 *                 (forward != backward, negative posterior, a term above LogZero) come back as DSR_E_CONSISTENCY
 *   prune         Lattice::prune(threshold) (:648-693): links with gamma > threshold leave the initial and intermediate nodes, unreachable
 *                 nodes are dropped and the rest renumbered in topological order
 *   prune_edges   Lattice::pruneEdges(edgesN) (:695-713): threshold = the edgesN-th smallest gamma over the links EdgeIterator sees
 *   purge         Lattice::purge() (:776-841): nodes from which no final node can be reached are dropped (links into them stay, as in the reference)
 *   get_state     per link: gamma, still on its node's list; per node: current index, still held by the lattice, forward and backward probability */
dsr_status dsr_lattice_read(const char* fileName, int noSelfLoops, int readData, dsr_lexicon* inputLex, dsr_lexicon* outputLex, dsr_lattice** out);
dsr_status dsr_lattice_rescore(dsr_lattice*, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX, float* score);
dsr_status dsr_lattice_best_hypo(const dsr_lattice*, int useInputSymbols, uint32_t* symbols, int cap, int* n);     /* symbols NULL: length only */
dsr_status dsr_lattice_gamma_probs(dsr_lattice*, double acScale, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX, double* logProb);
dsr_status dsr_lattice_prune(dsr_lattice*, double threshold);
dsr_status dsr_lattice_prune_edges(dsr_lattice*, unsigned edgesN);
dsr_status dsr_lattice_purge(dsr_lattice*);
dsr_status dsr_lattice_get_state(dsr_lattice*, double* gamma, int32_t* edgeLive, int32_t* nodeIndex, int32_t* nodeLive, double* forwardProb,
                                 double* backwardProb);              /* any pointer may be NULL */
/* 1-best writers over the rescoring tokens (call rescore first; DSR_E_CONSISTENCY when no final node holds a token -- the reference dereferences a
 * null token there).  fileName NULL or "": stdout, otherwise appended to.  Rows whose symbol equals endMarker are skipped.
 *   write_ctm        Lattice::writeCTM (lattice.cc:420-477): ";; utt cfrom score", "conv channel start duration word score" per output symbol
 *   write_phone_ctm  Lattice::writePhoneCTM (:479-537): the same per link, input symbols
 *   write_hypo_htk   Lattice::writeHypoHTK (:539-601): "utt.rec", a line per word (flag bit 0: times in 100 ns, bit 1: score), "."
 *   write_word_confs Lattice::writeWordConfs (:603-646): "uttId { {word} conf} ..." with conf = exp(-gamma) of the word's link */
dsr_status dsr_lattice_write_ctm(const dsr_lattice*, const dsr_lexicon* outputLex, const char* conv, const char* channel, const char* spk, const char* utt,
                                 double cfrom, double score, const char* fileName, double frameInterval, const char* endMarker);
dsr_status dsr_lattice_write_phone_ctm(const dsr_lattice*, const dsr_lexicon* inputLex, const char* conv, const char* channel, const char* spk,
                                       const char* utt, double cfrom, double score, const char* fileName, double frameInterval, const char* endMarker);
dsr_status dsr_lattice_write_hypo_htk(const dsr_lattice*, const dsr_lexicon* outputLex, const char* conv, const char* channel, const char* spk,
                                      const char* utt, double cfrom, double score, const char* fileName, int flag, double frameInterval,
                                      const char* endMarker);
dsr_status dsr_lattice_write_word_confs(const dsr_lattice*, const dsr_lexicon* outputLex, const char* fileName, const char* uttId, const char* endMarker);
/* flat image of a lattice for the gather across ranks (north star: "gather decoded lattices/1-best") */
size_t     dsr_lattice_pack_size(const dsr_lattice*);
dsr_status dsr_lattice_pack(const dsr_lattice*, void* buf, size_t bufBytes);
dsr_status dsr_lattice_unpack(const void* buf, size_t bytes, dsr_lattice** out);

/* debug/parity: per-frame token list (list order) of utterance 0 of the last decode with
   cfg.streams == 1 and dumpFrames enabled through dsr_decoder_enable_dump(). */
dsr_status dsr_decoder_enable_dump(dsr_decoder*, int enable);
dsr_status dsr_decoder_get_dump(dsr_decoder*, int64_t* nFrames, const int64_t** frameOff, const int32_t** node,
                                const float** ac, const float** lm, const int32_t** arc);

/* =====================================================================================
 * 6. Whole pipe: 8-ch analysis -> MVDR -> synthesis -> MFCC -> GMM -> Viterbi
 *    (the call sequence of SURVEY.md Appendix C.2-C.4 for a batch of utterances)
 * ===================================================================================== */
typedef struct dsr_pipe dsr_pipe;
dsr_status dsr_pipe_create(const dsr_fb* analysis, const dsr_fb* synthesis, dsr_bf* bf, dsr_mfcc* mfcc,
                           dsr_gmm* gmm, dsr_decoder* dec, int gmmMode, dsr_pipe** out);
void       dsr_pipe_destroy(dsr_pipe*);
/* x_dev [U][C][sampStride]; results as dsr_decoder_decode_batch.  Intermediates live in a workspace
   the pipe grows on demand (never inside a timed region after the first call of a given shape). */
dsr_status dsr_pipe_run(dsr_pipe*, const float* x_dev, const int32_t* nsamp_dev, const int32_t* nsamp_host,
                        int U, int C, int64_t sampStride, dsr_decode_result* res, int32_t* arcs_out,
                        uint32_t* words_out, int maxPath, void* stream);
/* The same in two halves (one batch in flight per pipe object): two pipes on two streams overlap the ragged end of
 * one batch's decode -- utterances finish at different times -- with the front end of the next batch. */
dsr_status dsr_pipe_submit(dsr_pipe*, const float* x_dev, const int32_t* nsamp_dev, const int32_t* nsamp_host,
                           int U, int C, int64_t sampStride, int maxPath, int want_paths, void* stream);
dsr_status dsr_pipe_collect(dsr_pipe*, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out);
/* fused = 1: analysis bank and beamformer run as one kernel when dsr_fb_analysis_beamform_supported() -- the channel snapshots are then never
 * written (intermediate 0 is not available, stage time 1 is zero and stage time 0 covers both); default 0 */
dsr_status dsr_pipe_set_fused(dsr_pipe*, int fused);
/* per-stage device time of the last run in milliseconds: [analysis, beamform, synthesis, mfcc, gmm, viterbi] */
dsr_status dsr_pipe_stage_ms(const dsr_pipe*, float ms[6]);
/* device pointers to the intermediates of the last run (borrowed): 0 X, 1 Y, 2 y, 3 feat, 4 scores */
dsr_status dsr_pipe_intermediate(const dsr_pipe*, int which, void** dev, int64_t* bytes);

/* PerfectReconstructionFFTAnalysisBank / PerfectReconstructionFFTSynthesisBank (btk/modulated/modulated.cc:686-970):
 * the 2M-band cosine-modulated pair; prototype of length 2M*m (analysis and synthesis objects each take their own).
 * analysis: x_dev [U][C][sampStride] -> X_dev [U][C][Tmax][2M] complex64, frames(nsamp) = ceil(nsamp/D) + 2m - 1;
 * synthesis: Y_dev [U][Tmax][2M] complex64, nframes_dev [U] -> y_dev [U][outStride] fp32, (nframes - (2m-1)) * D samples. */
typedef struct dsr_prfb dsr_prfb;
dsr_status dsr_prfb_create(const double* prototype, int M, int m, int r, dsr_prfb** out);
void       dsr_prfb_destroy(dsr_prfb*);
int        dsr_prfb_fft_len(const dsr_prfb*);
int        dsr_prfb_block_len(const dsr_prfb*);
int        dsr_prfb_analysis_frames(const dsr_prfb*, int nsamp);
int        dsr_prfb_synthesis_blocks(const dsr_prfb*, int nframes);
dsr_status dsr_prfb_analysis(const dsr_prfb*, const float* x_dev, const int32_t* nsamp_dev, int U, int C,
                             int64_t sampStride, int Tmax, float* X_dev, void* stream);
dsr_status dsr_prfb_synthesis(dsr_prfb*, const float* Y_dev, const int32_t* nframes_dev, int U, int Tmax,
                              int64_t outStride, float* y_dev, void* stream);

/* NormalFFTAnalysisBank (btk/modulated/modulated.cc:121-257) with getWindow (:72-97): windowed STFT, all M bins.
 * windowType 0 rectangle, 1 Hamming, 2 Hanning.  x_dev [U][C][sampStride] -> X_dev [U][C][Tmax][M] complex64;
 * frames(nsamp) = ceil(nsamp / D) + 1 (one zero-input frame, _processingDelay = 1), D = M >> r. */
typedef struct dsr_stft dsr_stft;
dsr_status dsr_stft_create(int M, int r, int windowType, dsr_stft** out);
void       dsr_stft_destroy(dsr_stft*);
int        dsr_stft_frames(const dsr_stft*, int nsamp);
int        dsr_stft_block_len(const dsr_stft*);
dsr_status dsr_stft_analysis(const dsr_stft*, const float* x_dev, const int32_t* nsamp_dev, int U, int C,
                             int64_t sampStride, int Tmax, float* X_dev, void* stream);

/* =====================================================================================
 * 6a. Zelinski post-filter on the beamformer output  (btk/postfilter/postfilter.cc:8-221,350-493:
 *     calcCSD, TimeAlignment, ZelinskiFilter_f, ZelinskiFilter, ZelinskiPostFilter; halfBandShift == false)
 *     type: 1 = Re(sum of CSDs), 2 = |sum| (the SWIG default), +8 = TYPE_ZELINSKI2 (the caller then passes wq() instead of
 *     arrayManifold() to set_manifold).  The densities start from scratch in every utterance (alpha = 0 for its first two
 *     frames, postfilter.cc:463-466); frames with frameX-1 < minFrames pass unfiltered (:471-473).
 *     X_dev [U][C][Tmax][M/2+1] complex64 (the analysis banks' snapshots), Y_dev [U][Tmax][M/2+1] complex64 (beamformer
 *     output) -> out_dev [U][Tmax][M/2+1]; wp1_dev (optional) [U][Tmax][M/2+1] fp32 = getPostFilterWeights().
 * ===================================================================================== */
typedef struct dsr_zelinski dsr_zelinski;
dsr_status dsr_zelinski_create(int fftLen, int chanN, double alpha, int type, int minFrames, dsr_zelinski** out);
void       dsr_zelinski_destroy(dsr_zelinski*);
dsr_status dsr_zelinski_set_manifold(dsr_zelinski*, int fbinX, const double* vec /* chanN complex128 */);   /* setArrayManifoldVector */
/* McCowanPostFilter (postfilter.cc:502-945): same handle type and apply; noise coherence per bin as in SubbandMVDR's setters */
dsr_status dsr_mccowan_create(int fftLen, int chanN, double alpha, int type, int minFrames, float threshold, dsr_zelinski** out);
dsr_status dsr_mccowan_set_noise_matrix(dsr_zelinski*, int fbinX, const double* Rnn /* [C][C] complex128 */);
dsr_status dsr_mccowan_set_diffuse_noise_model(dsr_zelinski*, const double* micPos /* [C][3] */, double sampleRate, double sspeed);
dsr_status dsr_mccowan_diagonal_loading(dsr_zelinski*, int fbinX /* < 0: all bins */, float diagonalWeight);
dsr_status dsr_mccowan_divide_nondiagonal(dsr_zelinski*, float myu);
/* LefkimmiatisPostFilter(output, fftLen, minSV, fbinX1, alpha, type, minFrames, threshold) (postfilter.h:180-202, postfilter.cc:948-1210):
 * the McCowan handle and setters; the pseudo-inverse of every bin's coherence matrix (singular values < minSV dropped) and
 * d^H pinv(R) d are computed inside apply when the matrices or the manifold have changed */
dsr_status dsr_lefkimmiatis_create(int fftLen, int chanN, double minSV, int fbinX1, double alpha, int type, int minFrames, float threshold, dsr_zelinski** out);
dsr_status dsr_zelinski_apply(dsr_zelinski*, const float* X_dev, const float* Y_dev, const int32_t* nframes_dev, int U, int Tmax,
                              float* out_dev, float* wp1_dev, void* stream);
/* The post-filter behind its beamformer: out = postfilter(X, bf(X)) -- ZelinskiPostFilter::setBeamformer (btk/postfilter/postfilter.h:100, postfilter.cc:376-384:
 * the filter takes snapshots and array manifold from the beamformer whose output it filters).  Where the filter streams the snapshots anyway (Zelinski on arrays
 * of other than 2/3/4/6/8 channels) the beamformer's sum is formed in the same pass over them; elsewhere the call is dsr_bf_apply_frames followed by
 * dsr_zelinski_apply.  Y_dev (optional, [U][Tmax][fftLen/2+1] complex64) receives bf(X). */
dsr_status dsr_zelinski_apply_bf(dsr_zelinski*, dsr_bf*, const float* X_dev, const int32_t* nframes_dev, int U, int Tmax,
                                 float* out_dev, float* wp1_dev, float* Y_dev, void* stream);
/* Carried densities for block-wise processing of long streams (BASELINE configs[4]).  The reference operator's auto/cross spectral densities
 * (postfilter.cc:428-497) live as long as the object: carry = 1 makes every apply of the same U continue the recursions where the previous call
 * stopped (stream u of one call = stream u of the next; the start-up alpha = 0 and minFrames count from a stream's own first frame);
 * reset_state begins new streams.  carry = 0 (default): every call is a batch of whole utterances.  chanN <= 64. */
dsr_status dsr_zelinski_carry(dsr_zelinski*, int on);
dsr_status dsr_zelinski_reset_state(dsr_zelinski*);

/* SubbandMMI (btk/beamformer/beamformer.h:264-312, beamformer.cc:1753-2319; beamformer.i:255-287): one generalized sidelobe canceller per
 * sound source; the output is the target source's GSC output, Zelinski post-filtered (pfType: postfilter.h:63-69 bits -- 0x01 real part /
 * 0x02 magnitude of the summed cross densities, 0x08 steer with the beamformer's own vector; 0 = no post-filter) and, after
 * use_binary_mask, zeroed (avgFactor < 0) or replaced by avgFactor x the recursive average of earlier outputs where another source's output
 * is stronger (type 0: the other sources' GSC outputs, 1: every source's upper-branch output).
 *   calc_weights      = calcWeights(sampleRate, delays[nSource][chanN])            one linear constraint per source (:1769-1780)
 *   calc_weights_n    = calcWeightsN(sampleRate, delays, NC)                       NC constraints: target + NC-1 nulls (:1788-1811)
 *   set_active_weights_f    = setActiveWeights_f(fbinX, packedWeights[rows][cols], option)      rows = nSource, cols = 2 (chanN - NC); option 1
 *                             resolves the scaling of the demixing matrix through its pseudo-inverse (:1821-1880)
 *   set_hi_active_weights_f = setHiActiveWeights_f(fbinX, pkdWa, pkdwb, option)    (:1891-1968)
 *   get: kind 0 wq [M][C], 1 wl [M][C], 2 B [M][C][C-NC], 3 array manifold [M][C], 4 wa [M][C-NC] of one source, complex128
 *   apply = next() for a batch: X_dev [U][chanN][Tmax][bins] complex64 snapshots -> Y_dev [U][Tmax][out_bins], bins = dsr_mmi_bins()
 *           (fftLen/2+1, or fftLen with halfBandShift); every utterance starts like a fresh object (frame counter, densities, average).
 *   pfType bit 0x04 = TYPE_APAB (beamformer.cc:2047-2049,2177-2179; ApabFilter postfilter.cc:225-340, channelX = chanN/2; it takes precedence
 *           over the Zelinski bits): the filter touches the bins below fftLen/2 only, so without halfBandShift the output frame is not
 *           conjugate-symmetric and ALL fftLen bins are handed over: out_bins = dsr_mmi_out_bins() = fftLen then (bins above fftLen/2 as the
 *           reference leaves them: conj of the unfiltered lower bin, or the mask's value); otherwise out_bins = bins.
 * Errors as the reference raises them: DSR_E_ERROR "call calcWeightsX() once" / wrong number of rows, DSR_E_DIMENSION for packed sizes and
 * bins. */
typedef struct dsr_mmi dsr_mmi;
dsr_status dsr_mmi_create(int fftLen, int chanN, int halfBandShift, int targetSourceX, int nSource, int pfType, double alpha, dsr_mmi** out);
void       dsr_mmi_destroy(dsr_mmi*);
int        dsr_mmi_bins(const dsr_mmi*);
int        dsr_mmi_out_bins(const dsr_mmi*);
int        dsr_mmi_chan_n(const dsr_mmi*);
int        dsr_mmi_fft_len(const dsr_mmi*);
dsr_status dsr_mmi_use_binary_mask(dsr_mmi*, double avgFactor, unsigned fwidth, unsigned type);
dsr_status dsr_mmi_calc_weights(dsr_mmi*, double sampleRate, const double* delays /*[nSource][chanN]*/);
dsr_status dsr_mmi_calc_weights_n(dsr_mmi*, double sampleRate, const double* delays /*[nSource][chanN]*/, unsigned NC);
dsr_status dsr_mmi_set_active_weights_f(dsr_mmi*, unsigned fbinX, const double* packedWeights, size_t rows, size_t cols, int option);
dsr_status dsr_mmi_set_hi_active_weights_f(dsr_mmi*, unsigned fbinX, const double* pkdWa, size_t nWa, const double* pkdwb, size_t nWb, int option);
dsr_status dsr_mmi_get(const dsr_mmi*, int srcX, int kind, double* out, size_t outDoubles);
dsr_status dsr_mmi_apply(dsr_mmi*, const float* X_dev, const int32_t* nframes_dev, int U, int Tmax, float* Y_dev, void* stream);

/* Single-channel WPE dereverberation of a subband sequence (SingleChannelWPEDereverberationFeature, btk/dereverberation/
 * dereverberation.cc:28-300; SWIG defaults iterationsN 2, loadDb -20, bandWidth 0, sampleRate 16000).  Y_dev [U][Nmax][M/2+1]
 * complex64 -> out_dev same shape; gn_dev (optional) [U][M/2+1][upperN-lowerN+1] complex128 = the prediction filters.  The
 * filters start from zero for every utterance (nextSpeaker() semantics). */
dsr_status dsr_wpe_single(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN,
                          int iterationsN, double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, void* stream);
/* the next utterance -- or the next block of a long stream -- of an object that was reset() but not nextSpeaker()-ed (dereverberation.cc:258-277:
 * reset() keeps _gn): gn_dev (required) holds the filters the call before left; they seed the first theta_n and are replaced by this call's */
dsr_status dsr_wpe_single_continue(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN,
                                   int iterationsN, double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, void* stream);
/* MultiChannelWPEDereverberation (dereverberation.h:89-157, dereverberation.cc:281-586): Y_dev [U][chanN][Nmax][fftLen/2+1] complex64 -> out_dev same shape;
 * gn_dev [U][chanN][fftLen/2+1][chanN*(upperN-lowerN+1)] complex128 (required).  filterChan < 0: own filter per channel; >= 0: all channels through that
 * channel's filter = the reference's getOutput when that channel's feature asks for the frame first (dereverberation.cc:381) */
dsr_status dsr_wpe_multi(const float* Y_dev, const int32_t* nframes_dev, int U, int chanN, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                         double loadDb, double bandWidth, double sampleRate, int filterChan, float* out_dev, double* gn_dev, void* stream);

/* =====================================================================================
 * 6b. LPC / MVDR spectral envelopes  (btk/feature/lpc.cc:44-207, lpc.h:134-195,291-331:
 *     WarpMVDRFeature, BurgMVDRFeature, WarpLPCFeature, BurgLPCFeature)
 *     method 0 = WarpFeature (warped autocorrelation + Levinson-Durbin), 1 = BurgFeature;
 *     kind 0 = MVDR envelope, 1 = LPC envelope.  frames_dev [T][dim] fp32 (the Hamming-windowed
 *     blocks) -> out_dev [T][dim/2+1] fp64.  order >= dim/2+1 => DSR_E_PARAMETER (lpc.h:126-127).
 * ===================================================================================== */
typedef struct dsr_lpc dsr_lpc;
dsr_status dsr_lpc_create(int dim, int order, int correlate, float warp, int method, int kind, dsr_lpc** out);
void       dsr_lpc_destroy(dsr_lpc*);
int        dsr_lpc_size(const dsr_lpc*);
dsr_status dsr_lpc_run(dsr_lpc*, const float* frames_dev, int64_t T, double* out_dev, void* stream);

/* =====================================================================================
 * 7. Stream/feature-operator API  (FeatureStream<Type,item>::next/reset/size/name/current/isEnd,
 *    btk/stream/stream.h:36-75).  Operators are reference counted handles that hold their
 *    upstream(s); next() returns a pointer to the operator's own output buffer (host memory),
 *    valid until the next call, exactly as the reference's _vector.
 * ===================================================================================== */
typedef struct dsr_stream dsr_stream;
enum { DSR_T_CHAR = 0, DSR_T_SHORT = 1, DSR_T_FLOAT = 2, DSR_T_DOUBLE = 3, DSR_T_COMPLEX = 4 };
dsr_status dsr_stream_next(dsr_stream*, int frameX /* -5 = next */, const void** data, size_t* n);
dsr_status dsr_stream_current(dsr_stream*, const void** data, size_t* n);
dsr_status dsr_stream_reset(dsr_stream*);
int          dsr_stream_size(const dsr_stream*);
int          dsr_stream_type(const dsr_stream*);
int          dsr_stream_frameX(const dsr_stream*);
int          dsr_stream_is_end(const dsr_stream*);
const char*  dsr_stream_name(const dsr_stream*);
void         dsr_stream_retain(dsr_stream*);
void         dsr_stream_release(dsr_stream*);
/* sources */
dsr_status dsr_sample_feature_create(int blockLen, int shiftLen, int padZeros, const char* name, dsr_stream** out);
dsr_status dsr_sample_feature_set_samples(dsr_stream*, const float* samples, size_t n, unsigned sampleRate);
/* SampleFeature::read(fn, format, samplerate, chX, chN, cfrom, to, outsamplerate, norm) (btk/feature/feature.cc:243-393; feature.i:487-489;
 * btk/src/superdirectiveBeamformer.cc:150-247 calls it from C++): RIFF/WAVE PCM of 8/16/24/32 bits; norm == 0 keeps the integer scale, otherwise
 * [-1, 1) x norm; chX is 1-based (0: DSR_E_CONSISTENCY "Multi-channel read is not yet supported."); an empty range or an unreadable file is DSR_E_IO;
 * sample-rate conversion is refused.  *nread = frames read; the stream is reset. */
dsr_status dsr_sample_feature_read(dsr_stream*, const char* fileName, int format, int samplerate, int chX, int chN, int cfrom, int to, int outsamplerate,
                                   float norm, int* nread);
int        dsr_sample_feature_sample_rate(const dsr_stream*);
/* PyFeatureStream equivalent (btk/stream/pyStream.h:44-130): a source whose frames the caller supplies;
   type is DSR_T_SHORT / DSR_T_FLOAT / DSR_T_DOUBLE / DSR_T_COMPLEX, data = nframes rows of `size` items */
dsr_status dsr_frame_source_create(int type, int size, const char* name, dsr_stream** out);
dsr_status dsr_frame_source_set_frames(dsr_stream*, const void* data, size_t nframes);
/* PyFeatureStream::reset() (pyStream.h:100-130) calls the Python object's reset() and iterates it afresh.  A reset() that reaches the source
   through a downstream operator marks its frames stale; `refill(user)` then runs before the next frame is served: it resets the caller's
   iterable and hands the new frames over with dsr_frame_source_set_frames; non-zero return = failure (DSR_E_PYTHON). */
dsr_status dsr_frame_source_set_refill(dsr_stream*, int (*refill)(void* user), void* user);
/* operators (ctor argument order as the reference headers) */
dsr_status dsr_analysis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r,
                                    int delayCompensationType, const char* name, dsr_stream** out);
dsr_status dsr_synthesis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r,
                                     int delayCompensationType, int gainFactor, const char* name, dsr_stream** out);
/* PerfectReconstructionFFTAnalysisBank(samp, prototype, M, m, r) / ...SynthesisBank(samp, prototype, M, m, r) (modulated.h:377-440) */
dsr_status dsr_pr_analysis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, const char* name, dsr_stream** out);
dsr_status dsr_pr_synthesis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, const char* name, dsr_stream** out);
/* NormalFFTAnalysisBank(samp, M, r, windowType) (modulated.i); samp delivers blocks of D = M >> r samples */
dsr_status dsr_normal_fft_bank_create(dsr_stream* samp, int M, int r, int windowType, const char* name, dsr_stream** out);
/* ZelinskiPostFilter(output, fftLen, alpha, type, minFrames) (postfilter.h:95-126): channels = the snapshot array's analysis
   streams (setSnapShotArray / setBeamformer), manifold = setArrayManifoldVector per bin */
dsr_status dsr_zelinski_stream_create(dsr_stream* output, int fftLen, double alpha, int type, int minFrames, const char* name, dsr_stream** out);
dsr_status dsr_zelinski_stream_set_channel(dsr_stream* pf, dsr_stream* chan);
/* McCowanPostFilter(output, fftLen, alpha, type, minFrames, threshold) (postfilter.i:113-126) on the same operator; its noise
   coherence setters: what 0 setNoiseSpatialSpectralMatrix(fbinX, data [C][C] complex), 1 setDiffuseNoiseModel(data = micPos [C][3],
   a = sampleRate, b = sspeed), 2 set(All)Level(s)OfDiagonalLoading(fbinX or -1, a), 3 divideAllNonDiagonalElements(a) */
dsr_status dsr_mccowan_stream_create(dsr_stream* output, int fftLen, double alpha, int type, int minFrames, float threshold, const char* name, dsr_stream** out);
dsr_status dsr_mccowan_stream_set_noise(dsr_stream* pf, int what, int fbinX, const double* data, int chanN, double a, double b);
/* highPassFilter(output, cutOffFreq, sampleRate) (postfilter.h:209-220, postfilter.cc:1222-1261): bins below fftLen*cutOffFreq/sampleRate are cut */
dsr_status dsr_highpass_filter_create(dsr_stream* output, float cutOffFreq, int sampleRate, const char* name, dsr_stream** out);
/* LefkimmiatisPostFilter (postfilter.i, postfilter.h:180-204) on the same operator and setters */
dsr_status dsr_lefkimmiatis_stream_create(dsr_stream* output, int fftLen, double minSV, int fbinX1, double alpha, int type, int minFrames, float threshold,
                                          const char* name, dsr_stream** out);
dsr_status dsr_zelinski_stream_set_manifold(dsr_stream* pf, int fbinX, const double* vec, int chanN);
/* SingleChannelWPEDereverberationFeature(samples, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate) (dereverberation.i:67-81) */
dsr_status dsr_wpe_single_stream_create(dsr_stream* samples, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth,
                                        double sampleRate, const char* name, dsr_stream** out);
/* MultiChannelWPEDereverberation + MultiChannelWPEDereverberationFeature(source, channelX) (dereverberation.i, dereverberation.h:89-174): one
 * operator per channel feature over the source's input streams (setInput order).  All channels of a frame go through the prediction filter
 * of the channel whose feature asks for the frame first (dereverberation.cc:381): by default the feature's own channel; the face that owns
 * the shared source sets the first asker with ..._set_filter_channel (< 0: every channel its own filter). */
dsr_status dsr_wpe_multi_feature_create(dsr_stream* const* channels, int channelsN, int channelX, int lowerN, int upperN, int iterationsN, double loadDb,
                                        double bandWidth, double sampleRate, const char* name, dsr_stream** out);
dsr_status dsr_wpe_multi_feature_set_filter_channel(dsr_stream* feature, int filterChan);
/* SubbandDS/GSC/MVDR as a stream: channels are analysis-bank streams (setChannel) */
dsr_status dsr_subband_bf_create(dsr_bf* weights, const char* name, dsr_stream** out);
dsr_status dsr_subband_bf_set_channel(dsr_stream* bf, dsr_stream* chan);
/* SubbandMMI as a stream (beamformer.i:255-287): channels through dsr_subband_bf_set_channel; frames beyond fftLen/2 are the conjugate mirror */
dsr_status dsr_subband_mmi_stream_create(dsr_mmi* weights, int fftLen, const char* name, dsr_stream** out);
/* SubbandOrthogonalizer(beamformer, outChanX) (beamformer.h:436-..., beamformer.cc:2817-2849) as a stream over a subband-beamformer operator:
 * outChanX <= 0: the beamformer's output; > 0: column outChanX-1 of the blocking matrices applied to the same snapshots (bins above M/2 as the
 * reference leaves them: the beamformer output's mirror) */
dsr_status dsr_subband_orthogonalizer_create(dsr_stream* beamformer, int outChanX, const char* name, dsr_stream** out);
dsr_status dsr_preemphasis_create(dsr_stream* samp, double mu, const char* name, dsr_stream** out);
dsr_status dsr_hamming_create(dsr_stream* samp, const char* name, dsr_stream** out);
dsr_status dsr_fft_create(dsr_stream* samp, int fftLen, const char* name, dsr_stream** out);
dsr_status dsr_spectral_power_create(dsr_stream* fft, int powN, const char* name, dsr_stream** out);
dsr_status dsr_vtln_create(dsr_stream* pow, int coeffN, double ratio, double edge, int version, const char* name, dsr_stream** out);
dsr_status dsr_mel_create(dsr_stream* mag, int powN, float rate, float low, float up, int filterN, int version, const char* name, dsr_stream** out);
dsr_status dsr_log_create(dsr_stream* mel, double m, double a, int sphinxFlooring, const char* name, dsr_stream** out);
dsr_status dsr_cepstral_create(dsr_stream* mel, int ncep, int type, const char* name, dsr_stream** out);
/* WarpMVDRFeature / BurgMVDRFeature (kind 0) and WarpLPCFeature / BurgLPCFeature (kind 1), lpc.h:115-128,280-293 */
dsr_status dsr_lpc_feature_create(dsr_stream* src, int order, int correlate, float warp, int method, int kind, const char* name, dsr_stream** out);
dsr_status dsr_storage_create(dsr_stream* src, const char* name, dsr_stream** out);
dsr_status dsr_mean_subtraction_create(dsr_stream* src, double devNormFactor, int runon, const char* name, dsr_stream** out);
/* the optional weight stream of MeanSubtractionFeature(src, weight, devNormFactor, runon) (feature.h, feature.cc:2577-2707): element 0 of its frames weighs
 * the frame in the batch statistics; in run-on mode frames with weight <= 0 do not update them */
dsr_status dsr_mean_subtraction_set_weight(dsr_stream* cmn, dsr_stream* weight);
dsr_status dsr_adjacent_create(dsr_stream* single, int delta, const char* name, dsr_stream** out);
dsr_status dsr_linear_transform_create(dsr_stream* src, int sz, const char* name, dsr_stream** out);
dsr_status dsr_linear_transform_set(dsr_stream*, const float* matrix /*[sz][srcSize]*/);
/* LinearTransformFeature::load(fileName, old) (feature.cc:2972-2976): gsl_matrix_float_load + the crop to (size x srcSize) */
dsr_status dsr_linear_transform_load(dsr_stream*, const char* fileName, int old);
/* StorageFeature::write(fileName, plainText) / read(fileName) (feature.cc:3025-3067), quirks kept (the count written is the index of the last
   frame; read() loads that many frames, one fewer than the file holds) */
dsr_status dsr_storage_write(dsr_stream*, const char* fileName, int plainText);
dsr_status dsr_storage_read(dsr_stream*, const char* fileName);

/* On-disk formats either side of the path, host side.
 * gsl_matrix_float_load / gsl_vector_float_load(m, fileName, old) (btk/matrix/gslmatrix.cc:27-96,133-240) into a caller matrix of rows x cols:
 *   old == 0: the GSL raw block (rows*cols native-endian floats); old != 0: Janus "FMAT"/"FVEC" (magic, big-endian sizes, a count that is
 *   skipped, big-endian floats; rows < 0 = derived from the file length).  Errors as the reference raises them: DSR_E_IO "Couldn't find magic
 *   number", "File empty", "Number of bytes in file = don't match matrix dimension"; DSR_E_DIMENSION "Cannot resize" when the file's matrix is
 *   larger than the caller's.  *rowsOut x *colsOut: the matrix size after the load (the file's, for old != 0). */
dsr_status dsr_fmat_load(const char* fileName, int old, int rows, int cols, float* data, int* rowsOut, int* colsOut);
dsr_status dsr_fmat_save(const char* fileName, int old, int rows, int cols, const float* data, int rowsUnset);
dsr_status dsr_fvec_load(const char* fileName, int old, int n, float* data, int* nOut);
dsr_status dsr_fvec_save(const char* fileName, int old, int n, const float* data);
/* HTK parameter files (btk/feature/feature.cc:4025-4318: ReadHTKHeader / WriteHTKHeader / ReadFloatBinary / WriteFloatBinary,
 * WriteHTKFeatureFile, HTKFeature): 12-byte header + float vectors of sampSize bytes.  isBigEndian is the reference's flag ("the machine is big
 * endian"): when 0 every field is byte-swapped, which gives the big-endian files HTK expects.  Compressed (_C) and CRC (_K) kinds: DSR_E_IO. */
dsr_status dsr_htk_write(const char* fileName, int nSamples, int sampPeriod, int sampSize, int parmKind, int isBigEndian, const float* data);
dsr_status dsr_htk_read_header(const char* fileName, int isBigEndian, int* nSamples, int* sampPeriod, int* sampSize, int* parmKind);
dsr_status dsr_htk_read(const char* fileName, int isBigEndian, float* data, size_t nFloats);

/* =====================================================================================
 * 8. The distribution set as the decoder sees it
 *    replaces Distrib::score(frameX) (asr/gaussian/distribBasic.h:48-50), DistribSet::find (:183-190), resetCache / resetFeature (:177-178) and
 *    the pull chain decoder -> distribution -> codebook -> feature stream (decoder.h:985, codebookBasic.cc:431-465): a GMM model bound to a
 *    feature-stream handle.  score() pulls frame frameX through the stream protocol (DSR_E_ITERATOR at the end, DSR_E_INDEX out of order), scores
 *    every distribution of that frame on the device once and serves the frame's other requests from that (the reference caches per codebook and
 *    frame).  decode_stream() = _Decoder::decode() for the utterance the stream holds: features, scores and token passing stay on the device.
 * ===================================================================================== */
typedef struct dsr_distribset dsr_distribset;
dsr_status dsr_distribset_create(dsr_gmm*, dsr_stream* feature, int gmmMode /* as dsr_gmm_score */, dsr_distribset** out);
void       dsr_distribset_destroy(dsr_distribset*);
int        dsr_distribset_ndists(const dsr_distribset*);
dsr_status dsr_distribset_find(const dsr_distribset*, const char* name, int* distX);
const char* dsr_distribset_name(const dsr_distribset*, int distX);
dsr_status dsr_distribset_score(dsr_distribset*, int distX, int frameX, float* score);
dsr_status dsr_distribset_reset_cache(dsr_distribset*);
dsr_status dsr_distribset_reset_feature(dsr_distribset*);
dsr_status dsr_decoder_decode_stream(dsr_decoder*, dsr_distribset*, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out, int maxPath);
/* Lattice::gammaProbsDist(dss, acScale, lmScale, lmPenalty, silPenalty, silSymbol) (asr/lattice/lattice.cc:331-341, 381-409): the acoustic score of
 * every link with an input symbol that leaves the initial node or a node of _nodes becomes the sum of its distribution's scores over the link's
 * frames (double accumulator, frame order; the frames of the bound feature stream are scored on the device, the sums are one gather kernel), then
 * gammaProbs.  DSR_E_INDEX: a link names a distribution or a frame the set / the stream does not have. */
dsr_status dsr_lattice_gamma_probs_dist(dsr_lattice*, dsr_distribset*, double acScale, double lmScale, double lmPenalty, double silPenalty,
                                        unsigned silenceX, double* logProb);

#ifdef __cplusplus
}
#endif
#endif /* DSR_H */
