/*
 * fakebob_hip_test.h -- test / profiling hooks of libfakebob_hip.so.
 *
 * Not part of the drop-in boundary (include/fakebob_hip.h): these entry points have no counterpart in the
 * reference.  tests/ use them to compare intermediate stages with the oracle (noise stream, int16 cast, MFCC,
 * compacted features), bench.py to time the dominant kernel and a run of NES iterations without host round
 * trips.  Same conventions as fakebob_hip.h (0 / negative FB_E_* + fb_last_error()).
 */
#ifndef FAKEBOB_HIP_TEST_H
#define FAKEBOB_HIP_TEST_H
#include "fakebob_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* z[half*N] float32 from the device Philox/Box-Muller (bit-exact contract) */
int fb_debug_noise(fb_engine *e, uint64_t seed, uint32_t iter, uint32_t stream,
                   int64_t N, int half, float *z);
/* the device int16 cast of model.score's float input (gmm_ubm_OSI.py:83-85) */
int fb_debug_quantize(fb_engine *e, const double *x, int64_t n, int bits_per_sample, int16_t *q);
/* front-end only: MFCC [T*num_ceps] of one utterance */
int fb_debug_mfcc(fb_engine *e, const int16_t *wav, int64_t n, float *mfcc, int *T);
/* compacted voiced CMVN'd features of one utterance: feats[Tv*dim] */
int fb_debug_feats(fb_engine *e, const int16_t *wav, int64_t n, float *feats,
                   int *Tv, int *T);

/* Which diagonal-GMM arithmetic the loaded model runs on: 2 = two-term f16 split (k_gmm_fx2w / k_gmm_fx2, default),
 * 1 = exact three-term bf16 split (k_gmm_bx3: chosen automatically when a parameter does not fit f16's exponent
 * range; FB_GMM_MODE=bx3 forces it).  Negative FB_E_* without a model.
 * (No reference counterpart: the reference runs Kaldi's float32 CPU code, gmm_ubm_kaldiHelper.py:202-221.) */
int fb_gmm_kernel_mode(fb_engine *e);
/* The kernel fb_score_* / the NES loop launch for the loaded GMM system: 1 = k_gmm_bx3, 2 = k_gmm_fx2 (any number of
 * variance groups, partial tiles, more than 28 models), 10 + P = k_gmm_fx2w (one variance group, 2 .. 28 models -- more than
 * 10 in two or three launches --: the
 * speaker models are scored as deltas from model 0 with 1 .. 3 partial products per K chunk (or class 6, below), chosen by fb_load_gmm PER
 * 32-COMPONENT TILE from how far the tile's components were adapted; P = the count most tiles run, *shift_rms
 * (nullable) returns the rms adaptation statistic; FB_GMM_DELTA_P forces one count for every tile; FB_GMM_NARROW=1
 * selects k_gmm_fx2 instead).  Negative FB_E_* without a model. */
int fb_gmm_kernel_variant(fb_engine *e, double *shift_rms);
/* k_gmm_fx2w's tile classes for the loaded model: how many component tiles run 1 / 2 / 3 partial products per K chunk
 * in their delta items (all zero, return value 0, when another kernel scores the model; 1 otherwise). */
int fb_gmm_delta_tiles(fb_engine *e, int *tiles_p1, int *tiles_p2, int *tiles_p3);
/* ... and how many run the F6 class: the leading f16 product plus the two correction products in block-scaled fp6
 * (fb_gmm_kernel_variant returns 16 when most tiles do; FB_GMM_DELTA_P=6 forces it for every tile, FB_GMM_DELTA_F6=0
 * keeps the rule from choosing it).  tiles_p1 + tiles_p2 + tiles_p3 + this = the model's component tiles. */
int fb_gmm_delta_tiles_f6(fb_engine *e);

/* the GMM kernel the engine scores with, on T rows of D features handed in as they are (no front-end): per-frame
 * log-likelihoods out[m * T + t] of every model (gmm-global-get-frame-likes without --average).  Lets the tests reach
 * inputs the front-end never produces: outliers, huge magnitudes, frames far from every component. */
int fb_debug_gmm_frames(fb_engine *e, const float *feats, int T, double *out);

/* number of UBM components that received posterior mass in the last i-vector batch (only their
 * rows of Sigma^-1 M / U are streamed by the contraction kernels) */
int fb_debug_iv_active(fb_engine *e, int *n_active);
/* gmm-gselect of the last i-vector batch: sel[rows * num_gselect] (nullable), info[5] = {which path ran -- 0: every
 * log-likelihood dumped + k_iv_select, 1: the threshold selection's general form (k_gmm_fx2_sel: lists of survivors), 2: its
 * wide form (k_gsel_w: 16-value records of the groups that reach the threshold) --, the flag (path 1: a list overflowed and
 * the dump redid the batch), most entries of one (row, chunk) list, entries in total, rows} */
int fb_debug_iv_gselect(fb_engine *e, int *sel, int64_t sel_cap, int64_t *info);
/* time `reps` back-to-back launches of the GMM log-likelihood kernel on the
 * engine's stream with HIP events over the current device feature buffer
 * (filled by the last score/get_grad call). ms_avg out. */
int fb_bench_gmm_kernel(fb_engine *e, int reps, double *ms_avg, int64_t *rows);
/* run `iters` NES iterations (get_grad + update, early stop disabled) on the
 * device without host round trips; returns elapsed ms (HIP events) and the
 * accumulated time of the GMM kernel alone (events around each launch when
 * time_gmm = 1; time_gmm = 2 with an i-vector system: around k_iv_solve_ll instead of the T-matrix contraction).  warmup < 0: continue the attack the previous call left on the
 * device -- nothing is uploaded or reset, `audio` is ignored (the timed region of
 * bench.py starts with its inputs resident in HBM). */
int fb_bench_nes(fb_engine *e, const fb_nes_params *p, const double *audio,
                 int64_t N, int warmup, int iters, int time_gmm,
                 double *ms_total, double *ms_gmm, int64_t *voiced_rows);

#ifdef __cplusplus
}
#endif
#endif
