/*
 * fakebob_hip.h -- C ABI of libfakebob_hip.so, the MI355X (gfx950) engine for
 * the FAKEBOB NES attack hot path.
 *
 * The reference has no FFI: its hot path crosses a *process* boundary
 * (subprocess.Popen of Kaldi programs).  Each entry point below replaces one
 * Python-level interface of the reference; the reference-side binding a
 * maintainer would add is a ctypes stub.
 *
 *   entry point               replaces (reference file:line)
 *   ------------------------  -------------------------------------------------
 *   fb_load_gmm               model_list / ubm paths handed to the wrappers
 *                             (gmm_ubm_OSI.py:15,45; attackMain.py:55,73-83)
 *   fb_set_frontend           pre-models/conf/{mfcc,vad}.conf + delta_opts read
 *                             by gmm_ubm_kaldiHelper.py:133,153,191
 *   fb_score_i16 / _f64       gmm_ubm_kaldiHelper.score (:270-291) and the
 *                             int16 cast of gmm_ubm_OSI.py:83-85
 *   fb_system_scores          wrapper post-processing gmm_ubm_OSI.py:89,
 *                             gmm_ubm_SV.py:77, gmm_ubm_CSI.py:93
 *   fb_get_grad               FakeBob.get_grad + loss_fn (FAKEBOB.py:223-299)
 *   fb_attack                 FakeBob.attack (FAKEBOB.py:139-221)
 *   fb_estimate_threshold     FakeBob.estimate_threshold (FAKEBOB.py:39-137)
 *   fb_get_grad_ext / fb_attack_ext   the same around a foreign `model` (README.md:136)
 *
 * Test / profiling hooks (no reference counterpart) live in fakebob_hip_test.h.
 *
 * Conventions: plain pointers and sizes, host memory unless a name ends in
 * _dev; every function returns 0 on success or a negative FB_E_* code and
 * leaves a message retrievable with fb_last_error() (the reference ignores
 * subprocess return codes, gmm_ubm_kaldiHelper.py:145-147 -- this is strictly
 * more).  One engine handle per (GPU, stream); a handle is not thread-safe,
 * different handles are independent.
 */
#ifndef FAKEBOB_HIP_H
#define FAKEBOB_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FB_OK 0
#define FB_E_ARG (-1)        /* bad argument / shape */
#define FB_E_HIP (-2)        /* HIP runtime error */
#define FB_E_STATE (-3)      /* model / frontend not loaded */
#define FB_E_NO_VOICED (-4)  /* an utterance has zero voiced frames (Kaldi
                                would silently drop it and mis-align rows) */
#define FB_E_NOMEM (-5)
#define FB_E_LIMIT (-6)      /* iteration cap reached (estimate_threshold) */
#define FB_E_CALLBACK (-7)   /* the score callback of a foreign model returned non-zero */

enum { FB_TASK_OSI = 0, FB_TASK_CSI = 1, FB_TASK_SV = 2 };
enum { FB_UNTARGETED = 0, FB_TARGETED = 1 };

typedef struct fb_engine fb_engine;

/* Kaldi front-end options ([EXT] conf/mfcc.conf, conf/vad.conf, delta_opts;
 * cmn flags fixed by gmm_ubm_kaldiHelper.py:196).  dither is always 0. */
typedef struct {
  double sample_freq;
  int frame_length;   /* samples */
  int frame_shift;    /* samples */
  int padded_length;  /* power of two: 512 */
  int num_mel_bins;
  int num_ceps;
  double low_freq, high_freq;
  double preemph;
  double cepstral_lifter;
  int snip_edges;
  int remove_dc;
  int use_energy;
  int raw_energy;
  double energy_floor;
  double vad_energy_threshold;
  double vad_energy_mean_scale;
  double vad_proportion_threshold;
  int vad_frames_context;
  int delta_window;
  int delta_order;
  int cmn_window;
  /* 0 (default): scores keep full precision.  1: emulate the text round trip of the reference -- Kaldi prints
   * every score as a float with 6 significant digits (`ark,t` / score files) and the helpers parse that text
   * (gmm_ubm_kaldiHelper.py:236-248, ivector_PLDA_kaldiHelper.py:310-338): raw scores are rounded to float32
   * and then to 6 significant decimal digits before any post-processing. */
  int text_scores;
  /* 0 (default): later stages read the MFCC matrix as computed.  1: emulate steps/make_mfcc.sh's default
   * `copy-feats --compress=true` (gmm_ubm_kaldiHelper.py:138-140, ivector_PLDA_kaldiHelper.py:163-165): the
   * matrix takes Kaldi's lossy CompressedMatrix round trip (8-bit codes between per-column 16-bit percentile
   * anchors; 16-bit codes for <= 8 frames) before VAD, deltas and CMVN read it. */
  int compress_feats;
  /* 0 (default): the MFCC arithmetic between Kaldi's float32 storage points is float64 (k_mfcc_r16).  1: Kaldi's own
   * precision -- BaseFloat = float32 end to end --, k_mfcc_f32: float32 window / FFT / mel /
   * DCT with one rounding per operation in a fixed order (the CPU oracle's twin is bit-identical), the frame's raw
   * log-energy C0 -- what compute-vad-decision votes on -- from the exact integer energy.  Needs the recipe's shape
   * (padded_length 512, raw_energy, <= 31 mel bins); fb_set_frontend refuses it otherwise. */
  int mfcc_f32;
} fb_frontend_cfg;

/* FakeBob hyper-parameters (FAKEBOB.py:21-37) + attack() arguments (:139) +
 * the RNG contract that replaces the unseeded np.random.normal (:234). */
typedef struct {
  int task;          /* FB_TASK_* */
  int attack_type;   /* FB_UNTARGETED / FB_TARGETED */
  double adver_thresh;
  double epsilon;
  int max_iter;
  double max_lr, min_lr;
  int samples_per_draw;
  double sigma;
  double momentum;
  int plateau_length;
  double plateau_drop;
  double threshold;  /* attack(threshold=...) */
  int target;        /* attack(target=...)  (index into speakers) */
  int true_label;    /* attack(true=...) */
  uint64_t seed;     /* Philox4x32-10 key */
  uint32_t stream;   /* Philox counter word 3 (utterance / attack id) */
  int bits_per_sample; /* attack(bits_per_sample=...): the int16 casts use 2^(bits_per_sample - 1) -- of every NES
                          sample before scoring (gmm_ubm_OSI.py:85) and of the returned audio (FAKEBOB.py:220).
                          2 .. 16; 0 means 16 */
} fb_nes_params;

const char *fb_last_error(void);
int fb_version(void);
int fb_device_count(void);

int fb_engine_create(int device, fb_engine **out);
int fb_engine_destroy(fb_engine *e);

void fb_default_frontend(fb_frontend_cfg *cfg);
int fb_set_frontend(fb_engine *e, const fb_frontend_cfg *cfg);

/* Diagonal GMMs in Kaldi DiagGmm internal form (float32): gconsts[M*C],
 * means_invvars[M*C*D], inv_vars[M*C*D].  Models whose inv_vars are bitwise
 * identical (mean-only MAP adaptation, build_spk_models.py:170) share the
 * quadratic term on device. */
int fb_load_gmm(fb_engine *e, int M, int C, int D, const float *gconsts,
                const float *means_invvars, const float *inv_vars);

/* i-vector / PLDA system: everything `sid/extract_ivectors.sh` + `ivector-plda-scoring` read
 * from pre-models/ (final.ubm, final.ie, mean.vec, transform.mat, plda) plus the enrolled
 * i-vectors the wrappers list in ivector.scp (ivector_PLDA_OSI.py:59-60,
 * ivector_PLDA_kaldiHelper.py:197-213,251-280).  Derived variables (diagonalised UBM for
 * gmm-gselect, Sigma^-1 M, U, PLDA-space enrolled vectors) are computed by the engine. */
typedef struct {
  int C, D, R, L, S;                /* Gaussians, feature dim, i-vector dim, LDA dim, speakers */
  int lda_cols;                     /* R, or R+1 when transform.mat carries an offset column */
  int num_gselect;                  /* 20 */
  double min_post;                  /* 0.025 */
  double prior_offset;              /* IvectorExtractor::prior_offset_ */
  const float *fg_weights;          /* [C]          FullGmm */
  const float *fg_means_invcovars;  /* [C*D] */
  const float *fg_inv_covars;       /* [C*D(D+1)/2] packed lower-triangular (SpMatrix) */
  const double *ie_M;               /* [C][D][R]    IvectorExtractor::M_ */
  const double *ie_sigma_inv;       /* [C][D(D+1)/2] IvectorExtractor::Sigma_inv_ (packed) */
  const float *mean_vec;            /* [R]          mean.vec */
  const float *lda;                 /* [L][lda_cols] transform.mat */
  const double *plda_mean;          /* [L] */
  const double *plda_transform;     /* [L][L] */
  const double *plda_psi;           /* [L] */
  const float *enrolled;            /* [S][R] enrolment i-vectors as stored by ivector-extract */
  const double *z_mean, *z_std;     /* [S] z-norm statistics of the speaker-model pickles */
} fb_ivector_system;

/* Loads an i-vector/PLDA system; scores are PLDA LLRs [B*S]; fb_system_scores / the NES
 * kernels apply (llr - z_mean) / z_std for every task (ivector_PLDA_OSI.py:119). */
int fb_load_ivector(fb_engine *e, const fb_ivector_system *sys, int task);

/* How raw per-model log-likelihoods become system scores:
 *  OSI / SV: model 0 is the UBM, S = M-1, score = raw[1+s] - raw[0]
 *  CSI     : S = M, score = (raw - z_mean) / z_std                          */
int fb_set_system(fb_engine *e, int task, const double *z_mean, const double *z_std);
int fb_num_speakers(fb_engine *e);

/* Average voiced-frame log-likelihood of every utterance under every model.
 * wav: concatenated samples, off[B+1] offsets; raw[B*M]; tv[B] (nullable) =
 * voiced frame counts. */
int fb_score_i16(fb_engine *e, const int16_t *wav, const int64_t *off, int B,
                 double *raw, int *tv);
int fb_score_f64(fb_engine *e, const double *audio, const int64_t *off, int B,
                 int bits_per_sample, double *raw, int *tv);
/* raw[B*M] -> scores[B*S] per fb_set_system */
int fb_system_scores(fb_engine *e, const double *raw, int B, double *scores);

/* One NES gradient estimate at `audio` (FAKEBOB.py:223-246).  noise_pos:
 * NULL -> Philox(seed, iter, stream); else float64 [N][spd/2] row-major
 * (the tensor np.random.normal would have produced).  grad[N] nullable,
 * score0[S]. */
int fb_get_grad(fb_engine *e, const fb_nes_params *p, const double *audio,
                int64_t N, uint32_t iter, const double *noise_pos,
                double *final_loss, double *grad, double *adver_loss,
                double *score0);

/* Whole attack loop on device.  noise_all NULL or [max_iter][N*(spd/2)].
 * adv_i16[N]; adver_f64[N] nullable; trace[max_iter*(3+S)] nullable rows of
 * {distance, adver_loss, lr_after_iter, score0[S]}; *n_trace rows written;
 * *success_flag = +1/-1 with the reference's iter < max_iter-1 rule (:219). */
int fb_attack(fb_engine *e, const fb_nes_params *p, const double *audio,
              int64_t N, const double *noise_all, int16_t *adv_i16,
              double *adver_f64, double *trace, int *n_trace,
              int *success_flag);

/* Seconds each iteration of the LAST fb_attack / fb_attack_ext took, iteration 0 from the attack's start: the
 * `used_time` column of the reference's trace pickle (FAKEBOB.py:205-212 times every loop body with time.time()).
 * Taken from the device's constant-rate clock where iteration i's loss is evaluated, so it is exact although the
 * host only looks at the device once per batch of iterations.  n <= the number of trace rows of that attack. */
int fb_attack_iter_seconds(fb_engine *e, double *seconds, int n);

/* Threshold sweep (FAKEBOB.py:39-137).  model_threshold: the system's own
 * threshold consulted by make_decisions.  Returns FB_E_LIMIT if
 * max_total_iters gradient steps did not reach acceptance. */
int fb_estimate_threshold(fb_engine *e, const fb_nes_params *p,
                          double model_threshold, const double *audio,
                          int64_t N, const double *noise_all,
                          int max_total_iters, double *score_out,
                          int *n_iters, int *n_outer, double *thr_final,
                          double *adver_f64);

/* ---- foreign models: the reference's plugin API ---------------------------------------------------
 * FakeBob takes ANY `model` object with score / make_decisions (README.md:136; FAKEBOB.py:53,89,250).  For a
 * model that is not one of this library's systems the per-iteration scores come from this callback --
 * audios[B][N] float64, utterance-major: row 0 = the current adversarial audio, rows 1.. = the antithetic NES
 * samples, exactly the columns FAKEBOB.py:234-238 hands to model.score; scores[B*S] out; return 0 -- and
 * everything else of the NES iteration (Philox noise, perturbation, loss, gradient estimate, momentum sign step,
 * clipping, loop control) still runs on the device.  No model has to be loaded into the engine; S = number of
 * enrolled speakers (1 for SV).  Arguments otherwise as fb_get_grad / fb_attack.
 * Lifetime: `audios` points into the engine's staging memory and is valid only DURING the call -- it is reused by the
 * next iteration; a model that keeps its batch must copy it (the Python mirror hands the model a copy). */
typedef int (*fb_score_cb)(void *ctx, const double *audios, int64_t N, int B, double *scores);
int fb_get_grad_ext(fb_engine *e, const fb_nes_params *p, int S, fb_score_cb cb, void *cb_ctx,
                    const double *audio, int64_t N, uint32_t iter, const double *noise_pos,
                    double *final_loss, double *grad, double *adver_loss, double *score0);
int fb_attack_ext(fb_engine *e, const fb_nes_params *p, int S, fb_score_cb cb, void *cb_ctx,
                  const double *audio, int64_t N, const double *noise_all, int16_t *adv_i16,
                  double *adver_f64, double *trace, int *n_trace, int *success_flag);

/* Launch chain of the device-controlled attack loop.  on = 1: 4 launches per NES iteration of a GMM system (MFCC; VAD +
 * deltas + CMVN; GMM log-likelihoods; finalisation + loss + loop control + update of iteration i + perturbation of
 * i + 1 in one) and, for i-vector systems, the five-workgroups-per-matrix posterior solve -- fastest for ONE or TWO
 * attacks per GPU, the reference's own use; on = 0: 6 launches -- finalisation and loss on their own, the front-end kernel at its own LDS size --, which interleave
 * better when several engines share a GPU (3 or more attacks in flight: +7 % NES iterations/s, profiles/r05_*;
 * FB_NO_FUSE=1: all 8); -1: default (fused unless
 * FB_NO_FUSE is set).  Trajectories are bit-identical either way. */
int fb_set_fused_chain(fb_engine *e, int on);

/* Work counters since engine creation: what the driver reduces over ranks next to the success counter
 * (attackMain.py:312,411 keeps success_cnt / total_cnt only). */
int fb_stats(fb_engine *e, int64_t *scored_utts, int64_t *scored_frames,
             int64_t *voiced_frames, int64_t *nes_iters);

/* ---- enrolment (build_spk_models.py) ------------------------------------------
 * fb_gmm_acc_stats replaces `gmm-global-acc-stats --update-flags=m final.dubm feats acc`
 * (build_spk_models.py:197-204): with the UBM loaded ALONE (fb_load_gmm, M = 1) it returns the float64
 * zeroth / first order statistics occ[C], F[C*D] of one utterance's voiced, CMVN'd delta features.  The
 * MAP mean update itself (gmm-global-est-map.cc:62-92, `MapDiagGmmUpdate`, means only) is a C*D
 * element-wise formula done by the host mirror (fakebob_amd/enroll.py).
 * fb_last_ivectors returns the i-vectors (B x R, float64, before mean subtraction / LDA) of the batch scored
 * last with an i-vector system: the enrolment identity of ivector_PLDA (build_spk_models.py:104-150). */
int fb_gmm_acc_stats(fb_engine *e, const int16_t *wav, int64_t n, double *occ, double *F, int *tv_out);
int fb_last_ivectors(fb_engine *e, int B, double *ivecs);

#ifdef __cplusplus
}
#endif
#endif
