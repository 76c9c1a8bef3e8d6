// fb_kernels.h -- host-callable launchers of the gfx950 kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fakebob_hip.h"
#include "../../include/fakebob_hip_test.h"

// ---- device-resident front-end tables (built on the host at fb_set_frontend)
struct FbFrontendDev {
  // scalars
  int L, P, shift, nb, nc, dim, order, dwin, cmn_window, snip_edges, remove_dc, use_energy,
      raw_energy, vad_ctx, mfcc_f32;
  double preemph, log_energy_floor;  // log_energy_floor = -inf when disabled
  double vad_thr, vad_mean_scale;
  float vad_prop;
  // tables (device pointers)
  const double *window;   // [L]   povey window (float32 values widened)
  const double *tw_half;  // [P/2][2] exp(-2 pi i m / (P/2))  (complex FFT of size P/2)
  const double *tw_full;  // [P/2+1][2] exp(-2 pi i k / P)    (real-FFT unpack)
  const int *mel_first;   // [nb]
  const int *mel_len;     // [nb]
  const int *mel_off;     // [nb] offset into mel_w
  const double *mel_w;    // packed weights (float32 values widened)
  const double *dct;      // [nc][nb] (float32 values widened)
  const double *lifter;   // [nc]
  const double *dscale;   // [(order+1)][2*order*dwin+1] delta kernels (float32 values widened)
  const float *f32_tab;   // k_mfcc_f32's tables as one float32 blob in its LDS layout (fb_mfcc_f32_table; null: not supported)
  const int *stop;        // nullable device flag: != 0 -> k_mfcc does nothing (attack already stopped)
  int mfcc_cus;           // k_mfcc_f32: compute units the launch may take (0 = all); set per batch by the engine
};

// ---- NES ----------------------------------------------------------------
// q[b][n] = int16((adver[n] + sigma*noise_b[n]) * 2^15), b in [0, 2*half]; column 0 is the
// un-noised adver.  noise: Philox(seed, iter, stream) or explicit float64 [N][half].
// dist_part[gridDim.x] gets per-block max |audio - adver| (audio nullable).
void fb_launch_perturb(hipStream_t s, const double *adver, const double *audio, int64_t N, int half,
                       double sigma, uint64_t seed, uint32_t iter, uint32_t stream,
                       const double *noise_pos, int16_t *q, double *dist_part, int *n_dist_part,
                       float *zbuf /* nullable: float32 normals [half][N] for the gradient kernel */,
                       const int *stop = nullptr /* nullable: device flag, != 0 -> the launch does nothing */,
                       int bits = 16 /* bits_per_sample of the cast: scale 2^(bits - 1) */);
void fb_launch_perturb_f64(hipStream_t s, const double *adver, const double *audio, int64_t N, int half,
                           double sigma, uint64_t seed, uint32_t iter, uint32_t stream, const double *noise_pos,
                           double *x, double *dist_part, int *n_dist_part, float *zbuf);
// *t = the device's constant-rate clock (wall_clock64) when the stream gets there
void fb_launch_stamp(hipStream_t s, unsigned long long *t);
// plain quantisation of float64 audio (model.score on float input)
void fb_launch_quantize(hipStream_t s, const double *x, int64_t n, int bits, int16_t *q);
// noise dump (tests)
void fb_launch_noise(hipStream_t s, uint64_t seed, uint32_t iter, uint32_t stream, int64_t N, int half,
                     float *z);

// Device-side control block of an attack (FAKEBOB.py:171-203): early stop on loss[0] < 0 (:181), the
// plateau learning-rate schedule (:195-200) and the per-iteration trace rows are handled by the loss
// kernel itself, so the host can queue several NES iterations ahead and only looks at the block once
// per batch.  Iterations queued behind the stopping one see `stop` and do nothing.
struct FbCtlDev {
  double lr, min_lr, plateau_drop;
  double *ls;  // [plateau_length] recent losses
  int n_ls, plateau_length;
  int stop, broke, stop_iter, iters_done, err, disable_stop;
  int pub_seq;   // the number of the last loss body that has published its results (k_gmm_finalize_loss_update: the
                 // update workgroups of the same launch poll it); the host counts the loss bodies it queues
  int pad_;
  unsigned long long *ticks;  // nullable: [1 + max_iter] device constant-rate clock (wall_clock64): [0] = start of the
                              // attack, [1 + it] = iteration it's loss evaluated -- the per-iteration times of the
                              // reference's trace (FAKEBOB.py:205-212)
};
struct FbNesDev {  // device control/result block of one NES iteration
  double adver_loss, final_loss, distance;
  int err;       // !=0: utterance (err-1) had no voiced frames
  int pad;
  double score0[62];
};
// scores + loss + summary (single block).  raw[B][M] -> scores[B][S] -> loss[B].
// znorm_all != 0 (i-vector systems): S = M and scores = (raw - z_mean)/z_std for every task.
void fb_launch_loss(hipStream_t s, const double *raw, const int *tv, int B, int M, int task, int znorm_all,
                    int attack_type, const double *z_mean, const double *z_std, double threshold,
                    double adver_thresh, int target, int true_label, const double *dist_part,
                    int n_dist_part, double *scores, double *loss, FbNesDev *out, FbCtlDev *ctl = nullptr,
                    double *trace = nullptr, int it = 0);
// k_grad_update (iteration `next_iter - 1`) + k_perturb (iteration next_iter) in one launch; device-controlled attacks
// with Philox noise and half <= FB_FUSE_MAX_HALF only.  Returns the number of distance partials written.
#define FB_FUSE_MAX_HALF 40
#define FB_FUSE_MAX_UPD_WG 192  // update workgroups k_gmm_finalize_loss_update may carry (N <= 49 152 samples; longer audio: k_update_perturb)
// the arguments of k_update_perturb, for the launch that carries it behind the GMM finalisation + loss (below)
struct FbUpdArgs {
  const double *loss;
  int64_t N;
  int half;
  double sigma;
  float *zbuf;
  double momentum, one_minus_m, epsilon;
  const double *audio;
  double *grad_m, *adver;
  uint64_t seed;
  uint32_t next_iter, stream;
  int16_t *q;
  double *dist_part;
  double qscale;
  unsigned long long *xch;  // nullable: B x M exchange slots of the finalising workgroups, every one FB_VAD_SENTINEL between launches
  int *role_ticket;         // nullable (FB_FIN_TICKET=1): k_gmm_finalize_loss_update's arrival ticket (zero between launches); null = roles by blockIdx
};
int fb_launch_update_perturb(hipStream_t s, const double *loss, int64_t N, int half, double sigma, float *zbuf,
                             double momentum, double one_minus_m, double epsilon, const double *audio, double *grad_m,
                             double *adver, const FbCtlDev *ctl, uint64_t seed, uint32_t next_iter, uint32_t stream,
                             int16_t *q, double *dist_part, int bits = 16);
// grad estimate (numpy-pairwise order) + optional momentum/sign/clip update.
// do_update: 0 = only grad_out; 1 = momentum+update with lr.
void fb_launch_grad_update(hipStream_t s, const double *loss, int64_t N, int half, double sigma,
                           const float *zbuf, const double *noise_pos, double *grad_out, int do_update,
                           double momentum, double one_minus_m, double lr, double epsilon,
                           const double *audio, double *grad_m, double *adver, const FbCtlDev *ctl = nullptr);

// ---- front-end ------------------------------------------------------------
// MFCC of every frame of a (ragged) batch.  wav_off[B+1], frame_off[B+1] device arrays.
// frame_rec: [total_frames][4] int32 = {absolute start sample (int64 in two words), start within the
// utterance, utterance length} (used by the P = 512 kernel instead of a search in frame_off)
void fb_launch_mfcc(hipStream_t s, const FbFrontendDev &fe, int melw_n, const int16_t *wav,
                    const int64_t *wav_off, const int *frame_off, const int32_t *frame_rec, int B,
                    int total_frames, float *mfcc);
// fb_frontend_cfg.mfcc_f32: the float32 kernel (frontend_f32_kernels.hip); false = this configuration is not one it takes
bool fb_mfcc_f32_supported(const FbFrontendDev &fe);
// the float32 table blob of k_mfcc_f32 (its LDS image up to the per-wave buffers) from the float64 host tables
#include <vector>
int fb_mfcc_f32_mel_pieces(const int *mel_len, int nb);  // chunks of 12 weights the filters split into (k_mfcc_f32: <= 64)
std::vector<float> fb_mfcc_f32_table(int L, int nb, int nc, const double *window, const double *tw_half, const double *tw_full,
                                     const int *mel_first, const int *mel_len, const int *mel_off, const double *mel_w, int melw_n,
                                     const double *dct, const double *lifter);
// uni_T > 0: every utterance has uni_n samples / uni_T frames and the first one starts at sample uni_base -- a frame's
// record is computed, not loaded (one dependent global round trip less at the head of every wave)
bool fb_launch_mfcc_f32(hipStream_t s, const FbFrontendDev &fe, int melw_n, const int16_t *wav, const int32_t *frame_rec,
                        int total_frames, float *mfcc, int uni_T = 0, int64_t uni_n = 0, int64_t uni_base = 0);
// VAD + per-utt voiced ranks.  vrank[f] = rank among voiced frames of its utt or -1; tv[b].
// counter: one device int, zero before the first launch (the kernel leaves it at zero); row_off[B+1]
// = exclusive scan of max(tv, 0), written by the workgroup that finishes last
// Kaldi CompressedMatrix round trip of every utterance's MFCC matrix, in place (t_max: longest utterance, frames)
void fb_launch_feat_compress(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, float *out, const int *frame_off, int B,
                             int t_max);  // out != mfcc: every workgroup reads the whole input matrix
void fb_launch_vad(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                   int *vrank, int *tv, int *counter, int *row_off);
// row_off[b] = sum_{b'<b} tv[b'] ; row_off[B] = total
// VAD + deltas + CMVN + voiced-row compaction of a batch whose utterances fit the CMVN window, one launch
// cm_out != nullptr (allowed when fb_vad_delta_cmvn_compresses(t_max)): the kernel also takes fb_launch_feat_compress's
// place -- the round trip on its LDS copy of the matrix, the compressed matrix written to cm_out (may be mfcc itself)
bool fb_vad_delta_cmvn_compresses(int t_max);
bool fb_launch_vad_delta_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                              int t_max, unsigned epoch, int *ticket, unsigned long long *pub, int *tv, int *row_off,
                              float *feats, float *cm_out);
// the same with every utterance split over FB_CMVN_PARTS (4) workgroups (no CompressedMatrix phase): part_sum = exchange
// slots, fb_vad_parts_doubles() 64-bit words, every 32-bit half FB_VAD_SENTINEL32 before the first launch; slot_set =
// launches of this kernel on the buffer since then (its two slot sets alternate with it)
#define FB_VAD_SENTINEL32 0x7ff87ff8u
#define FB_VAD_SENTINEL 0x7ff87ff87ff87ff8ull
size_t fb_vad_parts_doubles(const FbFrontendDev &fe, int B);
bool fb_launch_vad_delta_cmvn_p(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                                int t_max, unsigned epoch, int *ticket, unsigned long long *pub, int *tv, int *row_off,
                                float *feats, double *part_sum, unsigned slot_set,
                                bool spread = true /* pad the LDS request to one workgroup per CU (an attack alone on the GPU) */);
bool fb_launch_delta_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off,
                          const int *vrank, const int *row_off, int B, int t_max, float *feats);
// add-deltas (one workgroup per 32-frame chunk; chunk_off[B+1] = prefix of ceil(T_b/32)) + per-chunk
// column sums for the CMVN mean
void fb_launch_deltas(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off,
                      const int *chunk_off, int B, int total_chunks, float *dfeat, double *chunk_sum);
// apply-cmvn-sliding + select-voiced-frames -> compact feats[row][dim]
void fb_launch_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *dfeat, const int *frame_off,
                    const int *chunk_off, const double *chunk_sum, const int *vrank, const int *row_off, int B,
                    int total_chunks, bool any_long, float *feats);

// ---- diagonal GMM -----------------------------------------------------------
struct FbGmmDev {
  int M, C, D, n_tiles, n_items;      // items per tile = groups + models
  const int *item_model;              // [n_items]: -1 = quadratic (Q) item, else model index
  int item_model_host_q_first;        // 1: the item list is exactly {Q, model 0, 1, ..., M-1} (one variance group)
  // bf16x3 variant (k_gmm_bx3): K padded to 16*NK >= D + 3, images [n_tiles][n_items][3][NK][64] x 16 B
  int mode, NK;
  const unsigned int __attribute__((ext_vector_type(4))) * images_bx;
  // f16x2 variant (k_gmm_fx2): K padded to 16*NKF >= D + 1, images [n_tiles][n_items][2][NKF][64] x 16 B;
  // x^2 is scaled by 2^-sq_shift and the quadratic parameters by 2^+sq_shift
  int NKF;
  // power-of-two operand scalings (exact): frames x * 2^kx, x^2 * 2^kx2; the accumulators hold ll * 2^kacc
  int kx, kx2, kacc;
  const unsigned int __attribute__((ext_vector_type(4))) * images_fx;
  // k_gmm_fx2w (one variance group): images [n_tiles][1 + M][2][NKF][64] x 16 B of {Q, base model 0, delta_1 ..
  // delta_{M-1}}, delta_m = (means_invvars, gconst) of model m MINUS the base model's, all times log2 e and without
  // common power-of-two factor: every dimension is balanced by exact powers of two of its own, 2^kd / 2^kq, applied to
  // the frames in the kernel and inversely to the parameters here.  The COMPONENTS are stored sorted by how far the
  // other models moved them from the base model (the order is free under logsumexp), so that the partial products per K
  // chunk a delta item needs fall from tile to tile: tiles [0, delta_t3) are evaluated with 3, [delta_t6, delta_t2)
  // with 2, the rest with 1.  delta_p = the class most tiles use (1 .. 3, 6 = F6; 0: no delta images); anchor = the frames'
  // balancing factors and the components whose log2-likelihoods start the kernel's per-frame reference (layout:
  // fb_load_gmm)
  const unsigned int __attribute__((ext_vector_type(4))) * images_fd;
  int delta_p, delta_t3, delta_t2;
  int delta_t6;  // tiles [delta_t3, delta_t6): the F6 class (delta_t3 <= delta_t6 <= delta_t2; fb_load_gmm, gmm_wide_kernel.hip)
  // passes of k_gmm_fx2w (more than FB_FXW_MAX_M models): pass p scores the base model and the models pass_lo[p] ..
  // pass_lo[p + 1] - 1 from its own image buffer {Q, base, its deltas}; the launcher hands the kernel a copy of this
  // struct with images_fd = pass_images[p] and pass_first = pass_lo[p] (local model m >= 1 is model pass_first + m - 1 of
  // the M the partial sums are laid out for)
  int n_pass, pass_first;
  const unsigned int __attribute__((ext_vector_type(4))) * pass_images[3];
  int pass_lo[4];
  const float *anchor;
  const int *stop;  // nullable device flag: != 0 -> the launch does nothing (attack already stopped)
  const int *only_if;  // nullable device flag: == 0 -> the DUMP launch does nothing (the gselect rescue: fb_launch_gsel)
  int text_scores;  // fb_frontend_cfg.text_scores: raw scores through Kaldi's 6-significant-digit text output
  int fxw_sub;      // k_gmm_fx2w: component chunks ONE workgroup scores one after the other (0 / 1: one; 2: the launch of a GPU shared
                    // by three or more attacks -- half as many workgroups as compute units, the same partial sums bit for bit)
};
#define FB_FXW_MAX_PASS 3 // launches of k_gmm_fx2w per batch: 1 + 9 x 3 = 28 models (fb_load_gmm)
#define FB_FXW_MAX_M 10   // models ONE launch of k_gmm_fx2w takes: (1 + M) 10 KB items + the state of 2 M x 256 frames = 156 KB of LDS at M = 10
#define FB_FXW_ANCHORS 2  // anchor components of k_gmm_fx2w's per-frame reference (FbGmmDev::anchor)
#define FB_GMM_MODE_BX3 1
#define FB_GMM_MODE_FX2 2
#ifndef FB_FX_OCC
#define FB_FX_OCC 2  // k_gmm_fx2 workgroups per CU (launch bound and the launch's target block count)

#endif

// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is per DEVICE: a process that creates engines on several
// GPUs must repeat it on each.  One bit per device; the calls are idempotent, so a race between two host threads
// on the same device only repeats them.
#include <atomic>
static inline bool fb_device_needs_optin(std::atomic<unsigned long long> &mask, unsigned long long *bit) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  *bit = 1ull << (dev & 63);
  return (mask.load(std::memory_order_acquire) & *bit) == 0;
}


// true when fb_launch_gmm runs the one-wave-per-SIMD scoring kernel k_gmm_fx2w (256-frame strips, one round of <= 256
// workgroups): the engine sizes the component chunks for it
bool fb_gmm_use_wide(const FbGmmDev &g);
// gmm_wide_kernel.hip: the launch of k_gmm_fx2w (chunk c scores the component tiles c, c + n_chunks, ...); called by
// fb_launch_gmm
void fb_launch_gmm_wide(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap,
                        int n_chunks, float *part_m, float *part_s);
// part_m/part_s: [n_chunks][M][rows_pad]
void fb_launch_gmm(hipStream_t s, const FbGmmDev &g, const float *feats, const int *row_off_total,
                   int rows_cap, int n_chunks, float *part_m, float *part_s);
// single model (g.M == 1): ll[row][n_tiles*32] = every component log-likelihood (gmm-gselect input)
void fb_launch_gmm_dump(hipStream_t s, const FbGmmDev &g, const float *feats, const int *row_off_total,
                        int rows_cap, int n_chunks, float *ll);
// gmm-gselect WITHOUT the dump (round 6; k_gmm_fx2_sel / k_gsel_tau / k_gsel_final, gmm_kernels.hip): single model in
// the FX2 mode.  Workspace: gmax rows_cap x 2 n_tiles floats, tau rows_cap floats, glist rows_cap x n_chunks x cap 64-bit
// keys, gcnt rows_cap x n_chunks ints, flag 1 int (set when any row overflowed its lists: the caller then runs the dump +
// k_iv_select gated on it).  sel[row][nsel] as k_iv_select writes it.  fb_gsel_cap(n_chunks) = list entries per (row, chunk).
bool fb_gsel_applies(const FbGmmDev &g, int nsel, int n_chunks);
int fb_gsel_cap(int n_chunks);
int fb_gsel_chunks(int dump_chunks);   // the selection kernels' own chunk count (a power of two <= 8)
void fb_launch_gsel(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks,
                    int nsel, float *gmax, float *tau, unsigned long long *glist, int *gcnt, int *flag, int *sel);
// the wide form (k_gsel_w / k_gsel_final_w): fb_gsel_wide_chunks() > 0 = it applies, with that many component chunks.  gval:
// rows_cap x 32 n_tiles floats (the dump's buffer serves), gid: rows_cap x 2 n_tiles bytes, gcnt: rows_cap x n_chunks ints.
// No overflow and no rescue: sel[] is final (flag is raised by NaN features only).
int fb_gsel_wide_chunks(const FbGmmDev &g, int nsel, int rows_cap, int target_blocks = 256 /* workgroups the passes aim at */);
void fb_launch_gsel_wide(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks,
                         int nsel, float *gmax, float *tau, float *gval, unsigned char *gid, int *gcnt, int *flag, int *sel,
                         int *cnt /* nullable: fb_iv_bucket_cnt() -- the bucket sort's per-block counts, made here */, int Cpad);
// k_gmm_finalize + k_loss fused (GMM systems in the NES loop): counter = one int, zero before the first launch
void fb_launch_gmm_finalize_loss(hipStream_t s, const FbGmmDev &g, const float *part_m, const float *part_s,
                                 int rows_cap, int n_chunks, const int *row_off, int B, double *raw, int *counter,
                                 const int *tv, int task, int attack_type, const double *z_mean, const double *z_std,
                                 double threshold, double adver_thresh, int target, int true_label,
                                 const double *dist_part, int n_dist_part, double *scores, double *loss, FbNesDev *out,
                                 FbCtlDev *ctl, double *trace, int it, int pub_seq = 0, const FbUpdArgs *upd = nullptr);
// enrolment statistics of a single model from its dump matrix ll[rows][ld]: occ[C], F[C][D] (float64)
void fb_launch_gmm_post_stats(hipStream_t s, int C, int ld, int D, const float *ll, const float *feats,
                              const int *n_rows_ptr, int rows_cap, float *mx, float *inv_sum, double *occ,
                              double *F);
// raw[b][m] = mean over voiced rows of logsumexp (merging the chunk partials)
void fb_launch_gmm_finalize(hipStream_t s, const FbGmmDev &g, const float *part_m, const float *part_s,
                            int rows_cap, int n_chunks, const int *row_off, int B, double *raw);

// ---- i-vector / PLDA ------------------------------------------------------------
struct FbIvDev {
  int C, Cpad, D, R, L, S, lda_cols, nsel, triD, triR;
  float min_post;
  int text_scores;              // see fb_frontend_cfg
  double prior_offset;
  const float *fg_gconsts;      // [C]
  const float *fg_mic;          // [C][D]   means_invcovars
  const float *fg_P;            // [C][triD] inv_covars, packed lower-triangular
  const double *fg64;           // [C][triD + D + 1] float64 image for k_iv_fullcov_t: P with the diagonal halved,
                                //   then means_invcovars, then gconst
  const double *fgL;            // [C][FB_FCM_REC] (D = 72; else null): Cholesky image for k_iv_fullcov_mfma (ivector_kernels.hip)
  const unsigned char *tri_r, *tri_c;  // [triD] row / column of packed element e
  const double *sim;            // [C*D][R] Sigma^-1 M
  const double *u;              // [C][triR]
  const double *mean_vec;       // [R]
  const double *ldaT;           // [lda_cols][L]
  const double *plda_mean;      // [L]
  const double *pldaT;          // [L][L] transposed PLDA transform
  const double *plda_psi;       // [L]
  const double *train;          // [S][L] enrolled i-vectors in PLDA space
};
// What the posterior-solve kernels run BEHIND the solution of an utterance (round 5; fb_iv_tail.h): the back-end
// (ivector-subtract-global-mean | transform-vec | ivector-normalize-length, ivector-plda-scoring: k_iv_backend's body) in
// the workgroup that holds the solution, and -- inside the NES loop -- the loss / loop-control body of k_loss in the
// workgroup that finishes last (arrival counter, left at zero).  backend = 0: the solve kernels stop at the i-vectors
// and the caller launches k_iv_backend / k_loss itself (FB_IV_TAIL=split, shapes the tail does not take).
struct FbIvTail {
  int backend;          // 1: llr[b][s] written by the solving workgroup
  int loss;             // 1 (needs backend): the last arriver runs fb_loss_body<true, true>
  double *llr;          // [B][S]
  int *counter;         // arrivals (one int, zero before the first launch)
  const int *tv;
  int task, attack_type;
  const double *z_mean, *z_std;
  double threshold, adver_thresh;
  int target, true_label;
  const double *dist_part;
  int n_dist_part;
  double *scores, *loss_out;
  FbNesDev *out;
  FbCtlDev *ctl;
  double *trace;
  int it;
};
// doubles of LDS the tail needs behind the solve's own areas (back-end vectors + the loss body's two buffers)
size_t fb_iv_tail_lds_doubles(const FbIvDev &iv);
// true when the tail's loss part can run fused for a batch of B utterances (numpy's sum as one block: B - 1 <= 128)
bool fb_iv_tail_takes_loss(int B);
void fb_launch_iv_derive(hipStream_t s, int C, int D, int R, const double *M, const double *sinv_packed,
                         double *sim, double *u);
// bucket_ws: fb_iv_bucket_ws_ints() ints of workspace, zero before the first use; pairs / llf: rows_cap * nsel
size_t fb_iv_bucket_ws_ints(const FbIvDev &iv, int rows_cap);
// sel_gate: nullable device flag -- k_iv_select runs only when it is non-zero (sel[] then already holds fb_launch_gsel's
// selection; the gate is its overflow flag)
#define FB_IV_FB 64  // frames per partition block of the bucket sort (k_iv_bucket_count / _fill; k_gsel_final_w's workgroup)
int *fb_iv_bucket_cnt(const FbIvDev &iv, int *bucket_ws);   // where the per-block counts live inside bucket_ws
void fb_launch_iv_select_post(hipStream_t s, const FbIvDev &iv, const float *ll, const float *feats,
                              const int *n_rows_ptr, int rows_cap, int *sel, float *post, int *bucket_ws,
                              int *pairs, float *llf, const int *sel_gate = nullptr, bool run_select = true, bool run_count = true);
// gammaT [C][Bpad], XT [C*D][Bpad]: utterance-minor, zero-padded to Bpad (multiple of 32)
void fb_launch_iv_stats(hipStream_t s, const FbIvDev &iv, const float *feats, const int *row_off, const int *pairs,
                        const int *bucket_ws, const float *post, int B, int Bpad, double *gammaT, double *XT);
void fb_launch_iv_contract(hipStream_t s, const FbIvDev &iv, const double *gammaT, const double *XT, int B,
                           int Bpad, int n_kchunks, int *flags, int *active, int *n_active, double *linp,
                           double *quad, int *fail);
// ivector_solve.hip (k_iv_solve_ll): quad is consumed (factored in place); Aall = B x R right-hand sides + a row of
// R + 64 zeros behind them
void fb_launch_iv_solve_ll(hipStream_t s, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks,
                           int B, double *Aall, double *LinvAll, double *ivec, int *fail, const FbIvTail &tail);
// k_iv_solve_rw: the same system with the block rows of a matrix dealt over five workgroups (up-looking Cholesky); false
// = B x 5 workgroups are more than the chip holds at once (the caller runs fb_launch_iv_solve_ll).  LinvAll: TWO slot
// sets (fb_iv_solve_rw_linv_doubles), every 64-bit word 0x7ff87ff87ff87ff8 before the first launch and whenever `epoch`
// restarts; prog: fb_iv_solve_rw_prog_words() unsigned, zero then; ticket: one int, zero
size_t fb_iv_solve_rw_linv_doubles(const FbIvDev &iv, int B);
size_t fb_iv_solve_rw_prog_words(const FbIvDev &iv, int B);
bool fb_launch_iv_solve_rw(hipStream_t s, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks, int B,
                           double *Aall, double *LinvAll, double *ivec, int *fail, unsigned *prog, int *ticket, unsigned epoch,
                           const FbIvTail &tail);
void fb_launch_iv_backend(hipStream_t s, const FbIvDev &iv, const double *ivec, int B, double *llr);
