// smd_api.hip — the extern "C" boundary (include/smd_hotpath.h): argument validation, workspace carving, launches.
// No torch types, no allocation, no synchronisation; every launch goes to the caller's stream.
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "smd_kernels.h"
#include "smd_smooth_dev.h"   // block counts of the smoothness sweep / adjoint (host-side inline helpers)

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(hipError_t e, const char* what) {
  if (e != hipSuccess) return fail(SMD_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return SMD_OK;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct StripPlan { int rh, nsx, nsy; };
constexpr int kBwdAccRows = 16;    // = smd::kAccRows of smd_recon_bwd.hip
constexpr int kBwdMaskRows = 24;   // tallest backward strip: its per-row liveness masks (rows r0-3 .. r1+3) must fit 32 bits
constexpr int kMinStripRows = 4;   // lower bound of the rows-per-strip override; the workspace is sized for it
int max_strips(int h, int w, int cols) { return smd::ceil_div(w, cols)*smd::ceil_div(h, kMinStripRows); }


// Launch-shape knobs (smd_set_knob; unset = the built-in heuristics).  They select between partitions / code paths that must give the same
// results — the parity tests pin them to prove exactly that (tapered vs. plain partition, shared ring vs. per-wave loads, the two row loops
// of the backward) — and are NOT read from the environment: nothing on the call path calls getenv.  (A -DSMD_EXPERIMENTS build additionally
// seeds them from SMD_<NAME> environment variables once, for the scripts under scripts/dev.)
//   fwd_rh / bwd_rh rows per strip (>= 4); fwd_taper_b / bwd_taper_b samples at the end of the dispatch order that get short strips (0: none)
//   and fwd_taper_rh / bwd_taper_rh their height; fwd_ni supports per forward launch (1..4); fwd_share (default 1: with four scales a block
//   of the hot forward is the four scales of one strip and the target-side rows reach it through an LDS ring); bwd_skip (0 / 2: the
//   backward's row loop, overriding the SMD_BWD_SKIP_DEAD_ROWS flag of the call); bwd_wps (waves per strip of the backward);
//   bwd_guest_finalize; bwd_direct_level; loss_path_guests (0: the fused loss path launches its guest work as kernels of their own);
//   bwd_live (0: the backward ignores the forward's liveness table and runs every wave's row loop); bwd_scales_block (0: with one wave per strip a
//   block stays four strips of one scale instead of the four scales of one strip).
//   conv_two_tiles (1: the decoder's wide convolutions run two tiles of 32 output channels over one staged patch — eight waves per block — where the layer has
//   them; default 0: measured neutral, profiles/r06_conv_mfma_ablations.txt; same bits either way).
//   Experiments builds only: fwd_ahead (2: tap gathers two rows ahead, measured slower), bwd_pair (two supports per wave, dropped), smooth_chain.
struct KnobDef { const char* name; bool experiment; };
constexpr KnobDef kKnobs[] = {{"fwd_rh", false}, {"bwd_rh", false}, {"fwd_taper_b", false}, {"bwd_taper_b", false}, {"fwd_taper_rh", false},
                              {"bwd_taper_rh", false}, {"fwd_ni", false}, {"fwd_share", false}, {"bwd_skip", false}, {"bwd_wps", false},
                              {"bwd_guest_finalize", false}, {"bwd_direct_level", false}, {"loss_path_guests", false}, {"bwd_live", false}, {"bwd_scales_block", false}, {"conv_two_tiles", false},
                              {"fwd_ahead", true}, {"bwd_pair", true}, {"smooth_chain", true}};
constexpr int kNumKnobs = sizeof(kKnobs)/sizeof(kKnobs[0]);
constexpr int kKnobUnset = INT_MIN;
struct KnobTable {
  int v[kNumKnobs];
  KnobTable() {
    for (int i = 0; i < kNumKnobs; ++i) v[i] = kKnobUnset;
#ifdef SMD_EXPERIMENTS
    for (int i = 0; i < kNumKnobs; ++i) {   // once, at load time: SMD_FWD_RH=12 etc. for the dev scripts
      char env[64] = "SMD_";
      size_t k = 4;
      for (const char* c = kKnobs[i].name; *c && k + 1 < sizeof(env); ++c) env[k++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
      env[k] = 0;
      const char* e = getenv(env);
      if (e && *e) v[i] = atoi(e);
    }
#endif
  }
};
KnobTable g_knobs;
int knob_index(const char* name) {
  for (int i = 0; i < kNumKnobs; ++i) if (strcmp(kKnobs[i].name, name) == 0) return i;
  return -1;
}
int knob(const char* name, int dflt) {
  const int i = knob_index(name);
  return (i >= 0 && g_knobs.v[i] != kKnobUnset) ? g_knobs.v[i] : dflt;
}

StripPlan plan(int b, int S, int h, int w, int cols) {
  StripPlan p;
  p.rh = smd::pick_rows_per_strip(b, S, h, w, cols, 0);
  const int ov = knob(cols == smd::kFwdCols ? "fwd_rh" : "bwd_rh", 0);
  if (ov >= 1) p.rh = ov < kMinStripRows ? kMinStripRows : ov;
  if (cols == smd::kFwdCols && p.rh > 58) p.rh = 58;   // the K0-fused forward keeps the strip's rh + 3 + look-ahead row-table entries one per lane
  p.nsx = smd::ceil_div(w, cols);
  p.nsy = smd::ceil_div(h, p.rh);
  return p;
}

// Tapered partition of the fused kernels (smd_kernels.h: ReconMainArgs::b1): the last `b2` samples of the dispatch order get
// strips of `rh2` = rh/2 rows.  A launch is two to three "generations" of waves on the 4096 wave slots of the chip and a wave
// lives 30-40 us, so with equal units the last ones dispatched run almost alone for tens of microseconds (wave traces:
// scripts/dev/wave_trace.py; time-averaged occupancy 2.96 -> 3.19 waves per SIMD with the taper, forward 93 -> 87-89 us at
// cfg 2).  Default: a sixth of the batch, when the batch has at least four samples; the knobs *_taper_b / *_taper_rh override (b = 0: off).
void taper(int& b1, int& rh2, int& nsy2, int b, int h, const StripPlan& pl, const char* env_b, const char* env_rh) {
  int b2 = knob(env_b, -1), r2 = knob(env_rh, -1);
  if (b2 < 0) {
    b2 = (b >= 4) ? (b + 3)/6 : 0;
    // the taper buys the tail of a launch that is two or three generations of waves; a launch of five or more (384x640 at b = 12) pays for the short
    // strips' halo rows without needing it: half as many samples (backward 268 -> 262 us at cfg 4, 497 -> 485 at cfg 5; r04_fwd_shape_sweep.txt)
    if ((long)b*h*pl.nsx > 36000) b2 = (b2 + 1)/2;
  }
  if (b2 > b - 1) b2 = b - 1;
  if (r2 < kMinStripRows) r2 = pl.rh/2 < kMinStripRows ? kMinStripRows : pl.rh/2;
  b1 = b - b2; rh2 = r2; nsy2 = smd::ceil_div(h, r2);
}

// The forward's partition (shared with the backward, which reads the liveness table the forward indexed by ITS strips).
StripPlan fwd_partition(int b, int S, int h, int w, int& b1, int& rh2, int& nsy2) {
  const StripPlan pl = plan(b, S, h, w, smd::kFwdCols);
  taper(b1, rh2, nsy2, b, h, pl, "fwd_taper_b", "fwd_taper_rh");
  if (rh2 > 58) { rh2 = 58; nsy2 = smd::ceil_div(h, rh2); }   // as plan(): one row-table entry per lane
  return pl;
}

// ---- optional event-pair recording around the dominant kernels (bench.py roofline measurement) ----
struct ProfSlot { hipEvent_t* ev = nullptr; int cap = 0, used = 0; };
ProfSlot g_prof[5];   // SMD_PROF_*: dominant forward kernel, dominant backward kernel, whole forward entry point, whole backward entry point, prep launches

void prof_mark(int which, hipStream_t st, bool begin) {
  ProfSlot& p = g_prof[which];
  if (!p.ev || p.used >= p.cap) return;
  // a failed record only loses one timing sample (collect() reports the pairs that completed)
  if (begin) (void)hipEventRecord(p.ev[2*p.used], st);
  else { (void)hipEventRecord(p.ev[2*p.used + 1], st); ++p.used; }
}


// The K0-fused instantiations of k_recon_main exist for the SSIM loss and walk the low-resolution rows one at a time, which
// needs every pyramid level to be no taller than the image (always true for a decoder's outputs).
bool k0_fusable(const smd::ScaleSet& sc, int h, int flags) {
  if (flags & SMD_LOSS_L1) return false;
  for (int s = 0; s < sc.S; ++s) if (sc.hs[s] > h) return false;
  return true;
}

struct ReconWs { float* loss_partial; float* pose_partial; size_t bytes; };
// What the fused loss path adds to the reconstruction forward: the smoothness sweep (as guest blocks of the main launch, or launched behind
// it) and the in-launch weighted sum of the two losses.
struct LossPathFwd { smd::SmoothFwdJob job; smd::LossCombine comb; bool guests; };

ReconWs carve_recon(void* base, int b, int n, int S, int h, int w) {
  ReconWs r;
  size_t off = 0;
  char* p = (char*)base;
  r.loss_partial = (float*)(p + off); off += align256((size_t)S*b*max_strips(h, w, smd::kFwdCols)*sizeof(float));
  r.pose_partial = (float*)(p + off); off += align256((size_t)n*b*S*max_strips(h, w, smd::kBwdCols)*smd::kPoseSums*sizeof(float));
  r.bytes = off;
  return r;
}

unsigned* packed_live(float* packed, int b, int n, int h, int w) { return (unsigned*)(packed + smd::packed_live_offset_floats(b, n, h, w)); }
uint4* packed_rowtab(float* packed, int b, int n, int h, int w) { return (uint4*)(packed + smd::packed_rowtab_offset_floats(b, n, h, w)); }
unsigned* packed_arrive(float* packed, int b, int n, int h, int w) { return (unsigned*)(packed + smd::packed_arrive_offset_floats(b, n, h, w)); }

int check_dims(int b, int n, int S, int h, int w) {
  if (b < 1 || n < 1 || S < 1 || h < 2 || w < 2) return fail(SMD_E_INVALID, "invalid sizes b=%d n=%d S=%d h=%d w=%d (need h, w >= 2)", b, n, S, h, w);
  if (n > SMD_MAX_SUPPORTS) return fail(SMD_E_INVALID, "n=%d exceeds SMD_MAX_SUPPORTS=%d", n, SMD_MAX_SUPPORTS);
  if (S > 65535 || b > 65535) return fail(SMD_E_INVALID, "b or S exceeds the grid limit");
  if ((size_t)S*b*h*w >= ((size_t)1 << 32)) return fail(SMD_E_INVALID, "S*b*h*w must stay below 2^32");
  return SMD_OK;
}

int fill_scales(smd::ScaleSet& sc, const float* const* p, float* const* g, const int* hs, const int* ws, const int* keys, int S) {
  if (S < 1 || S > SMD_MAX_SCALES) return fail(SMD_E_INVALID, "S=%d outside [1, %d]", S, SMD_MAX_SCALES);
  if (!hs || !ws) return fail(SMD_E_INVALID, "null scale-size arrays");
  memset(&sc, 0, sizeof(sc));
  sc.S = S;
  for (int s = 0; s < S; ++s) {
    if (hs[s] < 1 || ws[s] < 1) return fail(SMD_E_INVALID, "scale %d has empty size %dx%d", s, hs[s], ws[s]);
    sc.p[s] = p ? p[s] : nullptr;
    sc.g[s] = g ? g[s] : nullptr;
    sc.hs[s] = hs[s]; sc.ws[s] = ws[s];
    sc.key[s] = keys ? keys[s] : s;
  }
  return SMD_OK;
}

}  // namespace

namespace smd {
char g_variant[2][128] = {"", ""};   // process-wide on purpose: autograd launches the backward from its own thread, the caller asks from another (a diagnostic label, last writer wins)
void note_variant(int which, const char* fmt, ...) {
  if (which < 0 || which > 1) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_variant[which], sizeof(g_variant[which]), fmt, ap);
  va_end(ap);
}
}  // namespace smd

extern "C" {

const char* smd_last_error(void) { return g_err; }
const char* smd_last_kernel_variant(int which) { return (which == 0 || which == 1) ? smd::g_variant[which] : ""; }
int smd_abi_version(void) { return SMD_ABI_VERSION; }

int smd_set_knob(const char* name, int value) {
  const int i = name ? knob_index(name) : -1;
  if (i < 0) return fail(SMD_E_INVALID, "unknown knob '%s'", name ? name : "(null)");
#ifndef SMD_EXPERIMENTS
  if (kKnobs[i].experiment) return fail(SMD_E_UNSUPPORTED, "knob '%s' exists in -DSMD_EXPERIMENTS builds only", name);
#endif
  g_knobs.v[i] = value;
  return SMD_OK;
}
void smd_reset_knobs(void) { for (int i = 0; i < kNumKnobs; ++i) g_knobs.v[i] = kKnobUnset; }

// ------------------------------------------------------------------------------------------------
int smd_disp_to_depth_fwd(const float* const* disp, const int* hs, const int* ws, int S, int b, int h, int w,
                          float min_depth, float max_depth, float* depth_up, float* disp_up, void* stream) {
  if (!disp || !depth_up) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1 || b > 65535) return fail(SMD_E_INVALID, "invalid sizes");
  if ((min_depth > 0.f || max_depth > 0.f) && !(min_depth > 0.f)) return fail(SMD_E_INVALID, "Min depth must be greater than 0. (%g)", min_depth);
  if (max_depth > 0.f && max_depth < min_depth) return fail(SMD_E_INVALID, "Max depth must be greater than min. (%g vs. %g)", max_depth, min_depth);
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, nullptr, hs, ws, nullptr, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s]) return fail(SMD_E_INVALID, "null disparity pointer for scale %d", s);
  return check_launch(smd::launch_disp_to_depth_fwd(sc, b, h, w, min_depth, max_depth, depth_up, disp_up, (hipStream_t)stream), "disp_to_depth_fwd");
}

size_t smd_disp_to_depth_workspace_bytes(const int* hs, const int* ws, int S, int b, int h, int w) {
  smd::ScaleSet sc;
  if (b < 1 || h < 1 || w < 1 || fill_scales(sc, nullptr, nullptr, hs, ws, nullptr, S)) return 0;
  return align256(smd::disp_to_depth_bwd_tmp_floats(sc, b, h, w, nullptr)*sizeof(float)) + 256;
}

int smd_disp_to_depth_bwd(const int* hs, const int* ws, int S, int b, int h, int w, float min_depth, float max_depth,
                          const float* depth_up, const float* g_depth_up, float* const* g_disp,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!depth_up || !g_depth_up || !g_disp || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1 || b > 65535) return fail(SMD_E_INVALID, "invalid sizes");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, nullptr, g_disp, hs, ws, nullptr, S)) return rc;
  for (int s = 0; s < S; ++s) if (!g_disp[s]) return fail(SMD_E_INVALID, "null gradient pointer for scale %d", s);
  const size_t need = smd_disp_to_depth_workspace_bytes(hs, ws, S, b, h, w);
  if (workspace_bytes < need) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  return check_launch(smd::launch_disp_to_depth_bwd(sc, b, h, w, min_depth, max_depth, depth_up, g_depth_up, (float*)workspace, false,
                                                    (hipStream_t)stream), "disp_to_depth_bwd");
}

// ------------------------------------------------------------------------------------------------
size_t smd_image_recon_workspace_bytes(int b, int n, int S, int h, int w) {
  if (b < 1 || n < 1 || S < 1 || h < 2 || w < 2) return 0;
  return carve_recon(nullptr, b, n, S, h, w).bytes;
}

size_t smd_packed_supports_bytes(int b, int n, int h, int w) {
  if (b < 1 || n < 1 || h < 2 || w < 2) return 0;
  return smd::packed_total_floats(b, n, h, w)*sizeof(float);
}

// The input-only half of the forward (k_recon_prep): everything that depends on the frames alone.  `sc` != null and K0 fusable:
// also the vertical up-sampling table of the pyramid.  Zeroes the arrival counters in the tail of `packed`.
static int recon_prep_impl(const float* tgt, const float* supp, float* supp_packed, const smd::ScaleSet* sc, int b, int n, int S, int h, int w,
                           int flags, hipStream_t st) {
  int kMaxPerPass = knob("fwd_ni", 4);   // supports held in registers by one launch (1..4)
  if (kMaxPerPass < 1 || kMaxPerPass > 4) kMaxPerPass = 4;
  smd::ReconPrepArgs p;
  memset(&p, 0, sizeof(p));
  p.tgt = tgt; p.supp = supp; p.packed = supp_packed;
  p.b = b; p.n = n; p.h = h; p.w = w; p.flags = flags & (SMD_USE_MIN | SMD_LOSS_L1 | SMD_USE_AUTOMASK);
  const StripPlan ipl = plan(b, 1, h, w, smd::kFwdCols);  // one "scale" only: shorter strips keep the chip full
  p.rh = ipl.rh; p.nsx = ipl.nsx; p.nsy = ipl.nsy;
  const bool fuse_k0 = sc && k0_fusable(*sc, h, flags);
  prof_mark(SMD_PROF_RECON_PREP, st, true);
  for (int i0 = 0; i0 < n; i0 += kMaxPerPass) {
    p.i0 = i0; p.ni = (n - i0 < kMaxPerPass) ? n - i0 : kMaxPerPass;
    p.first_pass = (i0 == 0); p.last_pass = (i0 + p.ni >= n);
    p.rowtab = (fuse_k0 && i0 == 0) ? packed_rowtab(supp_packed, b, n, h, w) : nullptr;
    p.arrive = (i0 == 0) ? packed_arrive(supp_packed, b, n, h, w) : nullptr;
    if (p.rowtab) { p.sc_S = S; for (int s = 0; s < S; ++s) { p.sc_hs[s] = sc->hs[s]; p.sc_ws[s] = sc->ws[s]; } }
    if (int rc = check_launch(smd::launch_recon_prep(p, st), "recon prep")) return rc;
  }
  prof_mark(SMD_PROF_RECON_PREP, st, false);
  return SMD_OK;
}

// Shared body of the two forward entry points.  `sc` != null: K0 fused — the first launch computes the depth from the
// low-resolution disparity pyramid and writes it to `depth_out`; otherwise the depth is read from `depth`.
static int recon_fwd_impl(const float* depth, float* depth_out, const smd::ScaleSet* sc, float min_depth, float max_depth,
                          const float* tgt, const float* supp, const float* T, const float* K,
                          const float* K_inv, const float* noise, uint64_t seed, float* supp_packed, float* err, uint8_t* sel, float* loss,
                          float* warp0, void* workspace, size_t workspace_bytes,
                          int b, int n, int S, int h, int w, int flags, void* stream, const LossPathFwd* lp = nullptr) {
  if (int rc = check_dims(b, n, S, h, w)) return rc;
  if ((!depth && !sc) || !tgt || !supp || !T || !K || !K_inv || !supp_packed || !sel || !loss || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (smd_packed_supports_bytes(b, n, h, w) >= ((size_t)1 << 32)) return fail(SMD_E_INVALID, "the packed buffer (%zu bytes) must stay below 2^32", smd_packed_supports_bytes(b, n, h, w));
  if ((size_t)n*b*3*h*w*4 >= ((size_t)1 << 32)) return fail(SMD_E_INVALID, "n*b*3*h*w*4 must stay below 2^32");
  if ((size_t)(h + 1)*(size_t)(w + 1) >= ((size_t)1 << 24)) return fail(SMD_E_INVALID, "(h+1)*(w+1) must stay below 2^24");
  ReconWs ws = carve_recon(workspace, b, n, S, h, w);
  if (workspace_bytes < ws.bytes) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = (hipStream_t)stream;
  prof_mark(SMD_PROF_RECON_FWD_ALL, st, true);
  int kMaxPerPass = knob("fwd_ni", 4);   // supports held in registers by one launch (1..4)
  if (kMaxPerPass < 1 || kMaxPerPass > 4) kMaxPerPass = 4;
  if (!err && n > kMaxPerPass) return fail(SMD_E_INVALID, "err may be NULL only when all %d supports fit one pass (%d)", n, kMaxPerPass);

  if (!(flags & SMD_PACKED_READY)) {   // once per sample: texel repack, target window sums, identity error (scale independent)
    if (int rc = recon_prep_impl(tgt, supp, supp_packed, sc, b, n, S, h, w, flags, st)) return rc;
  }

  smd::ReconMainArgs a;
  memset(&a, 0, sizeof(a));
  if (sc && !k0_fusable(*sc, h, flags)) {   // see k0_fusable: run the K0 kernel, then the reconstruction on its output
    if (int rc = check_launch(smd::launch_disp_to_depth_fwd(*sc, b, h, w, min_depth, max_depth, depth_out, nullptr, st), "disp_to_depth_fwd")) return rc;
    depth = depth_out; sc = nullptr;
  }
  if (sc) {
    a.sc = *sc; a.depth_out = depth_out; a.rowtab = packed_rowtab(supp_packed, b, n, h, w);
    a.a_scale = 1.f; a.a_off = 0.f;
    if (min_depth > 0.f || max_depth > 0.f) {  // to_scaled: i_max = 1/min, i_min = 1/max (0 if unset)
      const float i_max = 1.f/min_depth, i_min = max_depth > 0.f ? 1.f/max_depth : 0.f;
      a.a_scale = i_max - i_min; a.a_off = i_min;
    }
  }
  a.depth = depth; a.packed = supp_packed; a.T = T; a.K = K; a.Kinv = K_inv;
  a.noise = noise; a.err = err; a.sel = sel; a.partial = ws.loss_partial; a.warp0 = warp0;
  a.arrive = packed_arrive(supp_packed, b, n, h, w); a.loss = loss; a.loss_scale = 1.0/((double)S*b*h*w);
  a.live = packed_live(supp_packed, b, n, h, w);
  a.b = b; a.n = n; a.S = S; a.h = h; a.w = w; a.flags = flags;
  a.wscale = (float)((double)w/(double)(w - 1)); a.hscale = (float)((double)h/(double)(h - 1));
  a.inv_n = (float)(1.0/(double)n);
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  const StripPlan pl = fwd_partition(b, S, h, w, a.b1, a.rh2, a.nsy2);
  a.rh = pl.rh; a.nsx = pl.nsx; a.nsy = pl.nsy;
  if (lp) {
    a.comb = lp->comb;
    if (lp->guests) { a.sm = lp->job; a.guest_blocks = smd::smooth_main_blocks(a.sc, b); }
  }
  a.lookahead = knob("fwd_ahead", 1) == 2 ? 2 : 1;
  a.share = (knob("fwd_share", 1) != 0 && a.S == 4 && a.rh % 4 == 0 && (a.b1 >= a.b || a.rh2 % 4 == 0)) ? 1 : 0;
  for (int i0 = 0; i0 < n; i0 += kMaxPerPass) {
    a.i0 = i0; a.ni = (n - i0 < kMaxPerPass) ? n - i0 : kMaxPerPass;
    a.first_pass = (i0 == 0); a.last_pass = (i0 + a.ni >= n);
    if (i0 == 0) prof_mark(SMD_PROF_RECON_FWD, st, true);
    if (int rc = check_launch(smd::launch_recon_main(a, st), "image_recon_fwd")) return rc;
    if (a.last_pass) prof_mark(SMD_PROF_RECON_FWD, st, false);
    if (a.depth_out) { a.depth = a.depth_out; a.depth_out = nullptr; }   // later passes (n > 4) read the depth the first one wrote
  }
  if (lp && !lp->guests) { if (int rc = check_launch(smd::launch_smooth_main(a.sc, b, lp->job, st), "loss_path smoothness sweep")) return rc; }
  prof_mark(SMD_PROF_RECON_FWD_ALL, st, false);   // the loss is reduced inside the last launch (recon_main_reduce)
  return SMD_OK;
}

int smd_image_recon_supports_per_pass(void) { const int k = knob("fwd_ni", 4); return (k < 1 || k > 4) ? 4 : k; }

int smd_image_recon_prep(const float* tgt, const float* supp, float* supp_packed, const int* hs, const int* ws, int S,
                         int b, int n, int h, int w, int flags, void* stream) {
  if (int rc = check_dims(b, n, S > 0 ? S : 1, h, w)) return rc;
  if (!tgt || !supp || !supp_packed) return fail(SMD_E_INVALID, "null pointer");
  if (smd_packed_supports_bytes(b, n, h, w) >= ((size_t)1 << 32)) return fail(SMD_E_INVALID, "the packed buffer (%zu bytes) must stay below 2^32", smd_packed_supports_bytes(b, n, h, w));
  if ((size_t)n*b*3*h*w*4 >= ((size_t)1 << 32)) return fail(SMD_E_INVALID, "n*b*3*h*w*4 must stay below 2^32");
  smd::ScaleSet sc;
  const bool pyramid = hs && ws && S > 0;
  if (pyramid) { if (int rc = fill_scales(sc, nullptr, nullptr, hs, ws, nullptr, S)) return rc; }
  return recon_prep_impl(tgt, supp, supp_packed, pyramid ? &sc : nullptr, b, n, S, h, w, flags, (hipStream_t)stream);
}

int smd_image_recon_fwd(const float* depth, const float* tgt, const float* supp, const float* T, const float* K,
                        const float* K_inv, const float* noise, uint64_t seed, float* supp_packed, float* err, uint8_t* sel, float* loss,
                        float* warp0, void* workspace, size_t workspace_bytes,
                        int b, int n, int S, int h, int w, int flags, void* stream) {
  return recon_fwd_impl(depth, nullptr, nullptr, 0.f, 0.f, tgt, supp, T, K, K_inv, noise, seed, supp_packed, err, sel, loss, warp0, workspace, workspace_bytes,
                        b, n, S, h, w, flags, stream);
}

int smd_image_recon_disp_fwd(const float* const* disp, const int* hs, const int* ws, int S, float min_depth, float max_depth,
                             const float* tgt, const float* supp, const float* T, const float* K, const float* K_inv,
                             const float* noise, uint64_t seed, float* supp_packed, float* depth_up, float* err, uint8_t* sel, float* loss,
                             float* warp0, void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream) {
  if (!disp || !depth_up) return fail(SMD_E_INVALID, "null pointer");
  if ((min_depth > 0.f || max_depth > 0.f) && !(min_depth > 0.f)) return fail(SMD_E_INVALID, "Min depth must be greater than 0. (%g)", min_depth);
  if (max_depth > 0.f && max_depth < min_depth) return fail(SMD_E_INVALID, "Max depth must be greater than min. (%g vs. %g)", max_depth, min_depth);
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, nullptr, hs, ws, nullptr, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s]) return fail(SMD_E_INVALID, "null disparity pointer for scale %d", s);
  return recon_fwd_impl(nullptr, depth_up, &sc, min_depth, max_depth, tgt, supp, T, K, K_inv, noise, seed, supp_packed, err, sel, loss, warp0,
                        workspace, workspace_bytes, b, n, S, h, w, flags, stream);
}

static int recon_bwd_impl(const float* depth, float* supp_packed, const float* T, const float* K,
                          const float* K_inv, const uint8_t* sel, const float* g_loss, const float* g_in, float k0_scale,
                          float* g_depth, float* g_T, float* g_K, float* g_Kinv, void* workspace, size_t workspace_bytes,
                          int b, int n, int S, int h, int w, int flags, void* stream, smd::PoseFinJob* guest = nullptr,
                          float* g_direct = nullptr, int direct_scale = -1, float g_scale = 1.f) {
  if (int rc = check_dims(b, n, S, h, w)) return rc;
  if (!depth || !supp_packed || !T || !K || !K_inv || !sel || !g_loss || !g_depth || !g_T || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if ((flags & SMD_NEED_K_GRAD) && (!g_K || !g_Kinv)) return fail(SMD_E_INVALID, "SMD_NEED_K_GRAD requires g_K and g_Kinv");
  if (n >= SMD_SEL_MASKED) return fail(SMD_E_INVALID, "too many supports");
  ReconWs ws = carve_recon(workspace, b, n, S, h, w);
  if (workspace_bytes < ws.bytes) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = (hipStream_t)stream;

  smd::ReconBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.depth = depth; a.packed = supp_packed; a.T = T; a.K = K; a.Kinv = K_inv; a.sel = sel; a.g_loss = g_loss;
  a.g_depth = g_depth; a.pose_partial = ws.pose_partial;
  a.g_in = g_in; a.k0_scale = k0_scale; a.g_scale = g_scale;
  a.g_direct = g_direct; a.direct_scale = g_direct ? direct_scale : -1;
  a.arrive = packed_arrive(supp_packed, b, n, h, w) + 1;
  a.live = (n <= smd::kLiveSupports && knob("bwd_live", 1) != 0 && !(flags & SMD_BWD_NO_LIVE)) ? packed_live(supp_packed, b, n, h, w) : nullptr;   // (written by the forward on this buffer)
  { int fnsy2; const StripPlan fp = fwd_partition(b, S, h, w, a.fwd_b1, a.fwd_rh2, fnsy2); a.fwd_rh = fp.rh; }             // (same knobs as the forward call that filled it)
  a.g_T = g_T; a.g_K = (flags & SMD_NEED_K_GRAD) ? g_K : nullptr; a.g_Kinv = (flags & SMD_NEED_K_GRAD) ? g_Kinv : nullptr;
  a.b = b; a.n = n; a.S = S; a.h = h; a.w = w; a.flags = flags;
  a.wscale = (float)((double)w/(double)(w - 1)); a.hscale = (float)((double)h/(double)(h - 1));
  StripPlan pl = plan(b, S, h, w, smd::kBwdCols);
  if (pl.rh > kBwdMaskRows) { pl.rh = kBwdMaskRows; pl.nsy = smd::ceil_div(h, pl.rh); }            // a strip's row masks are 32-bit words (k_recon_bwd: rows r0-3 .. r1+3)
  if (n >= 2 && pl.rh > kBwdAccRows) { pl.rh = kBwdAccRows; pl.nsy = smd::ceil_div(h, pl.rh); }   // the supports' shares of dL/d depth are summed from LDS rows
  a.rh = pl.rh; a.nsx = pl.nsx; a.nsy = pl.nsy;
  taper(a.b1, a.rh2, a.nsy2, b, h, pl, "bwd_taper_b", "bwd_taper_rh");
  if (a.rh2 > kBwdMaskRows) { a.rh2 = kBwdMaskRows; a.nsy2 = smd::ceil_div(h, a.rh2); }
  if (n >= 2 && a.rh2 > kBwdAccRows) { a.rh2 = kBwdAccRows; a.nsy2 = smd::ceil_div(h, a.rh2); }
  a.pose_stride = S*pl.nsx*(a.nsy2 > pl.nsy ? a.nsy2 : pl.nsy);
  a.skip_level = knob("bwd_skip", (flags & SMD_BWD_SKIP_DEAD_ROWS) ? 2 : 0);
  // Waves per strip.  min(n, 4) (default): one support per wave — half as long work units (a launch is only ~2 generations of waves,
  // so its tail is a fraction of a unit) and the waves of a strip share the target-side rows through one L1; 1: a wave takes every
  // support of its strip in turn.  cfg 2, rocprofv3: 113 vs 115 us on coherent masks, 187 vs 212 us on incoherent inputs, 140 vs
  // 145 us in the bench (profiles/r03_ab_kernel_times.txt).
  // Round 5: ONE wave per strip, taking the supports in turn, once the launch has at least three generations of such waves (strips x scales x
  // samples >= 3 x 4096 slots: 384x640 at b = 12, 192x640 at b = 24).  With every support live the two shapes cost the same there (441.7 vs 445.5 us
  // at four supports, 243.2 vs 243.1 at two; 10 % in favour of one wave per support in launches half that size: r05_bwd_wps_sweep.txt); but a wave
  // that takes its supports in turn simply passes over the ones the liveness table calls dead, so its time follows the LIVE supports — whereas a
  // dead support's own wave frees an issue slot, not a place for another block (LDS-resident until the block's slowest wave ends).  cfg 5 on
  // the masks of a training run: 404 -> 362 us; three supports all live: 362 -> 346 (four waves' slots hold 4 strips instead of 1 1/3).
  const long units1 = (long)pl.nsx*pl.nsy*S*b;
  a.wps = knob("bwd_wps", (n >= 2 && units1 >= 3L*4096) ? 1 : (n < 4 ? n : 4));
  if (a.wps < 1) a.wps = 1;
  if (a.wps > 4) a.wps = 4;
  if (a.wps > n) a.wps = n;
  // four scales and two or four strips per block: blocks of the scales of a strip instead (k_recon_bwd) — every block must then have its strip
  const int spb = smd::kWavesPerBlock/a.wps;
  a.scales_block = (n > 1 && S == 4 && smd::kWavesPerBlock == 4 && (spb == 2 || spb == 4) && (pl.nsx*pl.nsy) % spb == 0
                    && (a.b1 >= b || (pl.nsx*a.nsy2) % spb == 0) && knob("bwd_scales_block", 1) != 0) ? 1 : 0;
#ifdef SMD_EXPERIMENTS
  // Two supports per wave (knob bwd_pair; experiment of round 4): the strips of a block must be those of the one-support-per-wave kernel
  // with wps = n, so that the K0-adjoint guest epilogue finds the same per-block pose entries.
  a.pair = (knob("bwd_pair", 0) != 0 && (n == 2 || n == 4) && (flags & SMD_USE_MIN) && a.skip_level == 0) ? 1 : 0;
  if (a.pair) a.wps = n;
#endif
  if (guest) {   // the caller's next launch finalises the pose sums: no in-launch hand-off (the kernel skips it when `arrive` is null)
    a.arrive = nullptr;
    const int spb = smd::kWavesPerBlock/a.wps;
    guest->a = a; guest->b1 = a.b1;   // (chain / sm of the job are the caller's: left as they are)
    guest->entries1 = S*smd::ceil_div(pl.nsx*pl.nsy, spb); guest->entries2 = S*smd::ceil_div(pl.nsx*a.nsy2, spb);
  }
  prof_mark(SMD_PROF_RECON_BWD_ALL, st, true);
  prof_mark(SMD_PROF_RECON_BWD, st, true);
  if (int rc = check_launch(smd::launch_recon_bwd(a, st), "image_recon_bwd")) return rc;
  prof_mark(SMD_PROF_RECON_BWD, st, false);
  prof_mark(SMD_PROF_RECON_BWD_ALL, st, false);   // g_T / g_K / g_Kinv come out of the same launch (pose_finalize_sample)
  return SMD_OK;
}

int smd_image_recon_bwd(const float* depth, const float* tgt, float* supp_packed, const float* T, const float* K,
                        const float* K_inv, const uint8_t* sel, const float* g_loss,
                        float* g_depth, float* g_T, float* g_K, float* g_Kinv, void* workspace, size_t workspace_bytes,
                        int b, int n, int S, int h, int w, int flags, void* stream) {
  (void)tgt;   // kept in the signature for ABI stability: the target is read from the packed buffer since ABI 3
  return recon_bwd_impl(depth, supp_packed, T, K, K_inv, sel, g_loss, nullptr, 0.f, g_depth, g_T, g_K, g_Kinv, workspace, workspace_bytes,
                        b, n, S, h, w, flags, stream);
}

size_t smd_image_recon_disp_workspace_bytes(const int* hs, const int* ws, int S, int b, int n, int h, int w) {
  const size_t base = smd_image_recon_workspace_bytes(b, n, S, h, w), k0 = smd_disp_to_depth_workspace_bytes(hs, ws, S, b, h, w);
  if (!base || !k0) return 0;
  return align256(base) + align256((size_t)S*b*h*w*sizeof(float)) + k0;
}

int smd_image_recon_disp_bwd(const int* hs, const int* ws, int S, float min_depth, float max_depth, const float* depth_up,
                             float* supp_packed, const float* T, const float* K, const float* K_inv, const uint8_t* sel,
                             const float* g_loss, const float* g_depth_up_in, float* const* g_disp, float* g_T, float* g_K, float* g_Kinv,
                             void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream) {
  if (!depth_up || !g_disp || !workspace) return fail(SMD_E_INVALID, "null pointer");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, nullptr, g_disp, hs, ws, nullptr, S)) return rc;
  for (int s = 0; s < S; ++s) if (!g_disp[s]) return fail(SMD_E_INVALID, "null gradient pointer for scale %d", s);
  const size_t need = smd_image_recon_disp_workspace_bytes(hs, ws, S, b, n, h, w);
  if (!need || workspace_bytes < need) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  const size_t base = align256(smd_image_recon_workspace_bytes(b, n, S, h, w));
  float* g_depth = (float*)((char*)workspace + base);
  float* k0_tmp = (float*)((char*)workspace + base + align256((size_t)S*b*h*w*sizeof(float)));
  float a_scale = 1.f;
  if (min_depth > 0.f || max_depth > 0.f) a_scale = 1.f/min_depth - (max_depth > 0.f ? 1.f/max_depth : 0.f);
  // The fused backward applies d depth / d(scaled disparity) itself (it has the depth in a register), so the K0 adjoint is a
  // pure resampling adjoint of g_depth: it neither reads depth_up again (24 MB at cfg 2) nor multiplies.
  smd::PoseFinJob guest;
  memset(&guest, 0, sizeof(guest));
  const bool ride = knob("bwd_guest_finalize", 1) != 0;   // 0: the in-launch hand-off of smd_image_recon_bwd instead
  // A pyramid level that already has the image size (normally level 0) needs no resampling adjoint: the fused backward stores its rows
  // straight into that level's gradient tensor and the K0 adjoint launches no blocks for it (one (b,h,w) read + write less; needs another
  // level to carry the launch the pose epilogue rides in).
  int direct = -1;
  if (S > 1 && knob("bwd_direct_level", 1) != 0)
    for (int s = 0; s < S && direct < 0; ++s) if (hs[s] == h && ws[s] == w) direct = s;
  if (int rc = recon_bwd_impl(depth_up, supp_packed, T, K, K_inv, sel, g_loss, g_depth_up_in, a_scale,
                              g_depth, g_T, g_K, g_Kinv, workspace, base, b, n, S, h, w, flags, stream, ride ? &guest : nullptr,
                              direct >= 0 ? g_disp[direct] : nullptr, direct)) return rc;
  return check_launch(smd::launch_disp_to_depth_bwd(sc, b, h, w, min_depth, max_depth, depth_up, g_depth, k0_tmp, true, (hipStream_t)stream,
                                                    ride ? &guest : nullptr, direct), "disp_to_depth_bwd");
}

// ------------------------------------------------------------------------------------------------
// Fused loss path (round 5): `forward_loss` of the kbr configuration (src/core/trainer.py:383-392, 436-437, 462-464) as ONE operator.
// Forward = the K0-fused reconstruction launch carrying the smoothness sweep as guest blocks and forming the weighted sum in-launch;
// backward = the fused reconstruction backward, the K0 adjoint's first launch carrying the pose epilogue (now through to the pose network's
// outputs) and the smoothness adjoint as guests, and the K0 adjoint's second launch adding into what the smoothness adjoint wrote.
static const char* loss_path_unsupported(const smd::ScaleSet& sc, int n, int h, int w, int flags, int* direct) {
  if (flags & SMD_LOSS_L1) return "loss_name 'l1'";
  if ((flags & SMD_USE_LAPLACIAN) || !(flags & SMD_USE_EDGES)) return "a smoothness term other than SmoothReg(use_edges=True)";
  int per_pass = knob("fwd_ni", 4);
  if (per_pass < 1 || per_pass > 4) per_pass = 4;
  if (n > per_pass) return "more supports than one forward pass holds";
  if (!k0_fusable(sc, h, flags)) return "a pyramid level taller than the image";
  if (sc.S < 2) return "a single pyramid level";
  int ident = 0, d = -1;
  for (int s = 0; s < sc.S; ++s) if (sc.hs[s] == h && sc.ws[s] == w) { ++ident; if (d < 0) d = s; }
  if (ident > 1) return "two pyramid levels of the image's size";
  if (ident == 1 && knob("bwd_direct_level", 1) == 0) return "knob bwd_direct_level = 0";
  if (knob("bwd_guest_finalize", 1) == 0) return "knob bwd_guest_finalize = 0";
  *direct = d;
  return nullptr;
}

size_t smd_loss_path_workspace_bytes(const int* hs, const int* ws, int S, int b, int n, int h, int w) {
  const size_t fwd = align256(smd_image_recon_workspace_bytes(b, n, S, h, w)) + smd_disp_smooth_workspace_bytes(hs, ws, S, b);
  const size_t bwd = smd_image_recon_disp_workspace_bytes(hs, ws, S, b, n, h, w);
  if (!smd_image_recon_workspace_bytes(b, n, S, h, w) || !smd_disp_smooth_workspace_bytes(hs, ws, S, b) || !bwd) return 0;
  return fwd > bwd ? fwd : bwd;
}

int smd_loss_path_fwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, float min_depth, float max_depth,
                      const float* tgt, const float* supp, const float* T, const float* K, const float* K_inv, uint64_t seed,
                      float* supp_packed, float* edge_weights, float* depth_up, uint8_t* sel, float* loss3, float* stats,
                      void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, float w_recon, float w_smooth, void* stream) {
  if (!disp || !depth_up || !edge_weights || !loss3 || !stats || !workspace || !tgt) return fail(SMD_E_INVALID, "null pointer");
  if ((min_depth > 0.f || max_depth > 0.f) && !(min_depth > 0.f)) return fail(SMD_E_INVALID, "Min depth must be greater than 0. (%g)", min_depth);
  if (max_depth > 0.f && max_depth < min_depth) return fail(SMD_E_INVALID, "Max depth must be greater than min. (%g vs. %g)", max_depth, min_depth);
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, nullptr, hs, ws, scale_keys, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s]) return fail(SMD_E_INVALID, "null disparity pointer for scale %d", s);
  int direct = -1;
  if (const char* why = loss_path_unsupported(sc, n, h, w, flags, &direct)) return fail(SMD_E_UNSUPPORTED, "the fused loss path does not serve %s", why);
  const size_t need = smd_loss_path_workspace_bytes(hs, ws, S, b, n, h, w);
  if (!need || workspace_bytes < need) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  if (!(flags & SMD_EDGES_READY)) {   // frame-only half of the smoothness term inline (also zeroes the sweep's and the combination's counters)
    if (int rc = check_launch(smd::launch_smooth_edges(sc, b, tgt, h, w, edge_weights, st), "disp_smooth_prep")) return rc;
  }
  const size_t base = align256(smd_image_recon_workspace_bytes(b, n, S, h, w));
  LossPathFwd lp;
  smd::smooth_fwd_job(sc, b, loss3 + 2, stats, (float*)((char*)workspace + base), edge_weights, &lp.job);
  lp.comb.out3 = loss3; lp.comb.arrive = lp.job.arrive + (size_t)S*b + 1; lp.comb.w_rec = w_recon; lp.comb.w_sm = w_smooth;
  lp.job.comb = lp.comb;
  lp.guests = knob("loss_path_guests", 1) != 0;
  const int rflags = flags & (SMD_USE_MIN | SMD_USE_AUTOMASK | SMD_PACKED_READY);
  return recon_fwd_impl(nullptr, depth_up, &sc, min_depth, max_depth, tgt, supp, T, K, K_inv, nullptr, seed, supp_packed, nullptr, sel, loss3 + 1, nullptr,
                        workspace, base, b, n, S, h, w, rflags, stream, &lp);
}

int smd_loss_path_bwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, float min_depth, float max_depth,
                      const float* depth_up, float* supp_packed, const float* T, const float* K, const float* K_inv, const uint8_t* sel,
                      const float* stats, const float* edge_weights, const float* g_loss, float w_recon, float w_smooth,
                      const float* aa, const float* t, const uint8_t* invert, const float* fs, const float* cs,
                      float* const* g_disp, float* g_T, float* g_K, float* g_Kinv, float* g_aa, float* g_t, float* g_fs, float* g_cs,
                      void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream) {
  if (!disp || !depth_up || !g_disp || !stats || !edge_weights || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if ((aa != nullptr) != (t != nullptr) || (aa != nullptr) != (g_aa != nullptr) || (aa != nullptr) != (g_t != nullptr)) return fail(SMD_E_INVALID, "aa, t, g_aa, g_t go together");
  if ((fs != nullptr) != (cs != nullptr) || (fs != nullptr) != (g_fs != nullptr) || (fs != nullptr) != (g_cs != nullptr)) return fail(SMD_E_INVALID, "fs, cs, g_fs, g_cs go together");
  if (fs && (!aa || !(flags & SMD_NEED_K_GRAD))) return fail(SMD_E_INVALID, "the intrinsics' chain rule needs the pose chain and SMD_NEED_K_GRAD");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, g_disp, hs, ws, scale_keys, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s] || !g_disp[s]) return fail(SMD_E_INVALID, "null pointer for scale %d", s);
  int direct = -1;
  if (const char* why = loss_path_unsupported(sc, n, h, w, flags, &direct)) return fail(SMD_E_UNSUPPORTED, "the fused loss path does not serve %s", why);
  const size_t need = smd_loss_path_workspace_bytes(hs, ws, S, b, n, h, w);
  if (!need || workspace_bytes < need) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  const size_t base = align256(smd_image_recon_workspace_bytes(b, n, S, h, w));
  float* g_depth = (float*)((char*)workspace + base);
  float* k0_tmp = (float*)((char*)workspace + base + align256((size_t)S*b*h*w*sizeof(float)));
  float a_scale = 1.f;
  if (min_depth > 0.f || max_depth > 0.f) a_scale = 1.f/min_depth - (max_depth > 0.f ? 1.f/max_depth : 0.f);
  const bool guests = knob("loss_path_guests", 1) != 0;
  smd::PoseFinJob job;
  memset(&job, 0, sizeof(job));
  job.chain.aa = aa; job.chain.t = t; job.chain.invert = invert; job.chain.g_aa = g_aa; job.chain.g_t = g_t;
  job.chain.fs = fs; job.chain.cs = cs; job.chain.g_fs = g_fs; job.chain.g_cs = g_cs; job.chain.h = h; job.chain.w = w;
  job.sm.stats = stats; job.sm.edge_w = edge_weights; job.sm.g_loss = g_loss; job.sm.g_scale = w_smooth; job.sm.accumulate_scale = direct;
  job.sm.blocks_per_sample = guests ? smd::smooth_bwd_blocks_per_sample(sc) : 0;
  const int rflags = flags & (SMD_USE_MIN | SMD_USE_AUTOMASK | SMD_NEED_K_GRAD | SMD_BWD_SKIP_DEAD_ROWS | SMD_BWD_NO_LIVE);
  if (int rc = recon_bwd_impl(depth_up, supp_packed, T, K, K_inv, sel, g_loss, nullptr, a_scale, g_depth, g_T, g_K, g_Kinv, workspace, base,
                              b, n, S, h, w, rflags, stream, &job, direct >= 0 ? g_disp[direct] : nullptr, direct, w_recon)) return rc;
  if (!guests) {   // the smoothness adjoint as a launch of its own, between the reconstruction backward (which wrote the direct level) and the K0 adjoint (which adds)
    if (int rc = check_launch(smd::launch_smooth_bwd(sc, b, nullptr, h, w, SMD_USE_EDGES, stats, g_loss, edge_weights, (hipStream_t)stream, w_smooth, direct), "loss_path smoothness adjoint")) return rc;
  }
  return check_launch(smd::launch_disp_to_depth_bwd(sc, b, h, w, min_depth, max_depth, depth_up, g_depth, k0_tmp, true, (hipStream_t)stream, &job, direct, true), "loss_path K0 adjoint");
}

// ------------------------------------------------------------------------------------------------
static int smooth_chunks(const int* hs, const int* ws, int S) {
  int mx = 1;
  for (int s = 0; s < S; ++s) {   // units of the streaming sweep, or blocks of 256 pixels of the second-order (laplacian) sweep
    const int u = smd::smooth_units_of(hs[s], ws[s]), p = smd::ceil_div(hs[s]*ws[s], 256);
    mx = (u > p ? u : p) > mx ? (u > p ? u : p) : mx;
  }
  return mx;
}

size_t smd_disp_smooth_workspace_bytes(const int* hs, const int* ws, int S, int b) {
  if (!hs || !ws || S < 1 || S > SMD_MAX_SCALES || b < 1) return 0;
  return align256((size_t)S*b*smooth_chunks(hs, ws, S)*2*sizeof(float) + (size_t)S*b*sizeof(double));   // per-unit partials + per-pair loss shares
}

size_t smd_disp_smooth_edge_weight_bytes(const int* hs, const int* ws, int S, int b) {
  smd::ScaleSet sc;
  if (b < 1 || fill_scales(sc, nullptr, nullptr, hs, ws, nullptr, S)) return 0;
  return smd::smooth_edge_bytes(sc, b);   // {wx, wy} per pixel of every level + the arrival counters of the sweep's in-launch second stage
}

int smd_gaussian_blur3x3(const float* x, float* out, int planes, int h, int w, int adjoint, void* stream) {
  if (!x || !out) return fail(SMD_E_INVALID, "null pointer");
  if (x == out) return fail(SMD_E_INVALID, "in-place blur is not supported");
  if (planes < 1) return fail(SMD_E_INVALID, "invalid sizes");
  if (h < 2 || w < 2) return fail(SMD_E_INVALID, "reflect padding by one needs at least 2 x 2 pixels (got %d x %d)", h, w);   // F.pad(mode='reflect') raises too
  return check_launch(smd::launch_blur3(x, out, planes, h, w, adjoint != 0, (hipStream_t)stream), "gaussian_blur3x3");
}

int smd_disp_smooth_prep(const float* img, const int* hs, const int* ws, int S, int b, int h, int w, int flags, float* edge_weights, void* stream) {
  if (!img || !edge_weights) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1 || b > 65535) return fail(SMD_E_INVALID, "invalid sizes");
  if (!(flags & SMD_USE_EDGES) || (flags & SMD_USE_LAPLACIAN)) return fail(SMD_E_INVALID, "smd_disp_smooth_prep serves SmoothReg(use_edges=True) without use_laplacian");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, nullptr, nullptr, hs, ws, nullptr, S)) return rc;
  return check_launch(smd::launch_smooth_edges(sc, b, img, h, w, edge_weights, (hipStream_t)stream), "disp_smooth_prep");
}

int smd_disp_smooth_fwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, int b,
                        const float* img, int h, int w, int flags, float* loss, float* stats, float* disp_grad, float* image_grad,
                        float* edge_weights, void* workspace, size_t workspace_bytes, void* stream) {
  if (!disp || !img || !loss || !stats || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1 || b > 65535) return fail(SMD_E_INVALID, "invalid sizes");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, nullptr, hs, ws, scale_keys, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s]) return fail(SMD_E_INVALID, "null disparity pointer for scale %d", s);
  const size_t need = smd_disp_smooth_workspace_bytes(hs, ws, S, b);
  if (workspace_bytes < need) return fail(SMD_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  if ((flags & SMD_USE_EDGES) && !edge_weights) return fail(SMD_E_INVALID, "SMD_USE_EDGES needs the edge_weights buffer (smd_disp_smooth_edge_weight_bytes)");
  if ((flags & SMD_EDGES_READY) && (!(flags & SMD_USE_EDGES) || (flags & SMD_USE_LAPLACIAN))) return fail(SMD_E_INVALID, "SMD_EDGES_READY goes with SMD_USE_EDGES (first-order form)");
  return check_launch(smd::launch_smooth_fwd(sc, b, img, h, w, flags, loss, stats, disp_grad, image_grad, (float*)workspace,
                                             edge_weights, (flags & SMD_EDGES_READY) != 0, (hipStream_t)stream), "disp_smooth_fwd");
}

int smd_disp_smooth_bwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, int b,
                        const float* img, int h, int w, int flags, const float* stats, const float* edge_weights, const float* g_loss,
                        float* const* g_disp, void* stream) {
  if (!disp || !img || !stats || !g_loss || !g_disp) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1 || b > 65535) return fail(SMD_E_INVALID, "invalid sizes");
  smd::ScaleSet sc;
  if (int rc = fill_scales(sc, disp, g_disp, hs, ws, scale_keys, S)) return rc;
  for (int s = 0; s < S; ++s) if (!disp[s] || !g_disp[s]) return fail(SMD_E_INVALID, "null pointer for scale %d", s);
  if ((flags & SMD_USE_LAPLACIAN) && (flags & SMD_USE_EDGES) && !edge_weights) return fail(SMD_E_INVALID, "the second-order adjoint needs the edge weights the forward cached");
  return check_launch(smd::launch_smooth_bwd(sc, b, img, h, w, flags, stats, g_loss, edge_weights, (hipStream_t)stream), "disp_smooth_bwd");
}

// ------------------------------------------------------------------------------------------------
// Un-fused operators
size_t smd_view_synth_workspace_bytes(int B, int h, int w) {
  if (B < 1 || h < 2 || w < 2) return 0;
  return align256((size_t)B*smd::ceil_div(h*w, 256)*smd::kPoseSums*sizeof(float));
}

int smd_view_synth_fwd(const float* input, const float* depth, const float* T, const float* K, const float* K_inv,
                       float* warp, float* depth_warp, uint8_t* mask_valid, int B, int C, int h, int w, void* stream) {
  if (!input || !depth || !T || !K || !K_inv || !warp) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || B > 65535 || C < 1 || h < 2 || w < 2) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  return check_launch(smd::launch_view_synth_fwd(input, depth, T, K, K_inv, warp, depth_warp, mask_valid, B, C, h, w, (hipStream_t)stream), "view_synth_fwd");
}

int smd_view_synth_bwd(const float* input, const float* depth, const float* T, const float* K, const float* K_inv,
                       const float* g_warp, const float* g_depth_warp, float* g_input, float* g_depth, float* g_T, float* g_K, float* g_Kinv,
                       void* workspace, size_t workspace_bytes, int B, int C, int h, int w, void* stream) {
  if (!input || !depth || !T || !K || !K_inv || !g_warp || !g_depth || !g_T || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || B > 65535 || C < 1 || h < 2 || w < 2) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  if (workspace_bytes < smd_view_synth_workspace_bytes(B, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_view_synth_bwd(input, depth, T, K, K_inv, g_warp, g_depth_warp, g_input, g_depth, g_T, g_K, g_Kinv,
                                                 (float*)workspace, B, C, h, w, (hipStream_t)stream), "view_synth_bwd");
}

size_t smd_photo_error_workspace_bytes(int N, int C, int h, int w) {
  if (N < 1 || C < 1 || h < 2 || w < 2) return 0;
  return align256((size_t)N*3*C*h*w*sizeof(float));
}

int smd_photo_error_fwd(const float* pred, const float* target, float* err, int N, int C, int h, int w, int flags, float weight_ssim, void* stream) {
  if (!pred || !target || !err) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || N > 65535 || C < 1 || h < 2 || w < 2) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d h=%d w=%d", N, C, h, w);
  if (!(weight_ssim >= 0.f && weight_ssim <= 1.f)) return fail(SMD_E_INVALID, "Invalid SSIM weight. (%g vs. [0, 1])", weight_ssim);
  return check_launch(smd::launch_photo_error_fwd(pred, target, err, N, C, h, w, flags, weight_ssim, (hipStream_t)stream), "photo_error_fwd");
}

int smd_photo_error_bwd(const float* pred, const float* target, const float* g_err, float* g_pred, void* workspace, size_t workspace_bytes,
                        int N, int C, int h, int w, int flags, float weight_ssim, void* stream) {
  if (!pred || !target || !g_err || !g_pred || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || N > 65535 || C < 1 || h < 2 || w < 2) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d h=%d w=%d", N, C, h, w);
  if (workspace_bytes < smd_photo_error_workspace_bytes(N, C, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  if (!(weight_ssim >= 0.f && weight_ssim <= 1.f)) return fail(SMD_E_INVALID, "Invalid SSIM weight. (%g vs. [0, 1])", weight_ssim);
  return check_launch(smd::launch_photo_error_bwd(pred, target, g_err, g_pred, (float*)workspace, N, C, h, w, flags, weight_ssim, (hipStream_t)stream), "photo_error_bwd");
}

// ------------------------------------------------------------------------------------------------
// RegressionLoss
size_t smd_regression_workspace_bytes(size_t N) {
  if (N < 1) return 0;
  return align256((size_t)smd::regr_blocks(N)*3*sizeof(float));
}

int smd_regression_fwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* loss, float* err, float* stats,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!pred || !target || !loss || !stats || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || N > ((size_t)1 << 40)) return fail(SMD_E_INVALID, "invalid size N=%zu", N);
  if (workspace_bytes < smd_regression_workspace_bytes(N)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_regression_fwd(pred, target, mask, N, flags, loss, err, stats, (float*)workspace, (hipStream_t)stream), "regression_fwd");
}

int smd_regression_bwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* stats, const float* g_loss,
                       float* g_pred, float* g_target, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pred || !target || !stats || !g_loss || !workspace || (!g_pred && !g_target)) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || N > ((size_t)1 << 40)) return fail(SMD_E_INVALID, "invalid size N=%zu", N);
  if (workspace_bytes < smd_regression_workspace_bytes(N)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_regression_bwd(pred, target, mask, N, flags, stats, g_loss, g_pred, g_target, (float*)workspace, (hipStream_t)stream),
                      "regression_bwd");
}

size_t smd_recon_reduce_workspace_bytes(int B, int h, int w) {
  if (B < 1 || h < 1 || w < 1) return 0;
  return align256((size_t)smd::ceil_div(B*h*w, 256)*sizeof(float));
}

int smd_recon_reduce_fwd(const float* err_warp, const float* err_static, const float* mask, const float* noise, uint64_t seed, float* err, uint8_t* sel,
                         float* loss, void* workspace, size_t workspace_bytes, int n, int B, int h, int w, int flags, void* stream) {
  if (!err_warp || !err || !sel || !loss || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if ((flags & (SMD_MASK_EXPLAINABILITY | SMD_MASK_UNCERTAINTY)) && !mask) return fail(SMD_E_INVALID, "Must provide a 'mask' when masking...");
  if ((flags & SMD_USE_AUTOMASK) && !err_static) return fail(SMD_E_INVALID, "Must provide the original 'source' images when automasking...");
  if (n < 1 || n >= SMD_SEL_MASKED || B < 1 || h < 1 || w < 1) return fail(SMD_E_INVALID, "invalid sizes");
  if (workspace_bytes < smd_recon_reduce_workspace_bytes(B, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_recon_reduce_fwd(err_warp, err_static, mask, noise, seed, err, sel, loss, (float*)workspace, n, B, h, w, flags,
                                                   (hipStream_t)stream), "recon_reduce_fwd");
}

int smd_recon_reduce_bwd(const uint8_t* sel, const float* g_loss, float* g_err_warp, const float* err_warp, const float* err_static,
                         const float* mask, float* g_mask, int n, int B, int h, int w, int flags, void* stream) {
  if (!sel || !g_loss || !g_err_warp) return fail(SMD_E_INVALID, "null pointer");
  if (n < 1 || B < 1 || h < 1 || w < 1) return fail(SMD_E_INVALID, "invalid sizes");
  if (mask && (!g_mask || !err_warp || ((flags & SMD_USE_AUTOMASK) && !err_static))) return fail(SMD_E_INVALID, "a masked reduction needs err_warp, err_static (with automask) and g_mask");
  return check_launch(smd::launch_recon_reduce_bwd(sel, g_loss, g_err_warp, err_warp, err_static, mask, g_mask, n, B, h, w, flags, (hipStream_t)stream), "recon_reduce_bwd");
}

// ------------------------------------------------------------------------------------------------
// Monodepth decoder glue
static bool dec_sizes_ok(long long planes, int h, int w) {
  return planes >= 1 && h >= 2 && w >= 2 && (long long)(h + 2)*(w + 2) < (1ll << 30) && planes*(((long long)(h + 2)*(w + 2) + 255)/256) < (1ll << 31);
}
size_t smd_decoder_glue_workspace_bytes(int B, int C, int h, int w) {
  if (B < 1 || C < 1 || h < 1 || w < 1) return 0;
  return align256(smd::decoder_bias_partials(B, C, h, w)*sizeof(float));
}
int smd_elu_pad_fwd(const void* x, const float* bias, void* out, int B, int C, int h, int w, int apply_elu, int dtypes, void* stream) {
  if (!x || !out) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || C < 1 || !dec_sizes_ok((long long)B*C, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  return check_launch(smd::launch_elu_pad_fwd(x, bias, out, B, C, h, w, apply_elu, dtypes, (hipStream_t)stream), "elu_pad_fwd");
}
int smd_elu_pad_bwd(const void* x, const float* bias, const void* g_out, void* g_x, float* g_bias, void* workspace, size_t workspace_bytes,
                    int B, int C, int h, int w, int apply_elu, int dtypes, void* stream) {
  if (!x || !g_out || !g_x || (g_bias && !workspace)) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || C < 1 || !dec_sizes_ok((long long)B*C, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  if (g_bias && workspace_bytes < smd_decoder_glue_workspace_bytes(B, C, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_elu_pad_bwd(x, bias, g_out, g_x, g_bias, (float*)workspace, B, C, h, w, apply_elu, dtypes, (hipStream_t)stream), "elu_pad_bwd");
}
static bool head_sizes_ok(int B, int C, int h, int w) {
  return B >= 1 && C >= 1 && C < 65535 && dec_sizes_ok((long long)B*C, h, w) && (long long)B*((h + 47)/48) < 65536 && h < 65536 && (long long)B*C*(h + 2)*(w + 2) < (1ll << 40);
}
size_t smd_conv3x3_head_workspace_bytes(int B, int C, int h, int w) {
  if (!head_sizes_ok(B, C, h, w)) return 0;
  return align256(smd::conv_head_partials(B, C, h, w)*sizeof(float));
}
int smd_conv3x3_head_fwd(const void* xp, const float* weight, const float* bias, float* y, int B, int C, int h, int w, int act, void* stream) {
  if (!xp || !weight || !y) return fail(SMD_E_INVALID, "null pointer");
  if (!head_sizes_ok(B, C, h, w) || (act & ~(1 | SMD_HEAD_X_BF16))) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d act=%d", B, C, h, w, act);
  return check_launch(smd::launch_conv_head_fwd(xp, (act & SMD_HEAD_X_BF16) != 0, weight, bias, y, B, C, h, w, act & 1, (hipStream_t)stream), "conv3x3_head_fwd");
}
int smd_conv3x3_head_bwd(const void* xp, const float* weight, const float* y, const float* g_y, void* g_xp, float* g_weight, float* g_bias,
                         void* workspace, size_t workspace_bytes, int B, int C, int h, int w, int act, void* stream) {
  if (!weight || !y || !g_y || (!g_xp && !g_weight) || (g_weight && (!xp || !workspace)) || (g_bias && !g_weight)) return fail(SMD_E_INVALID, "null pointer");
  if (!head_sizes_ok(B, C, h, w) || (act & ~(1 | SMD_HEAD_X_BF16))) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d act=%d", B, C, h, w, act);
  if (g_weight && workspace_bytes < smd_conv3x3_head_workspace_bytes(B, C, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_conv_head_bwd(xp, (act & SMD_HEAD_X_BF16) != 0, weight, y, g_y, g_xp, g_weight, g_bias, (float*)workspace, B, C, h, w, act & 1, (hipStream_t)stream), "conv3x3_head_bwd");
}
static bool thin_sizes_ok(int B, int C, int h, int w) { return (C == 16 || C == 32) && head_sizes_ok(B, C, h, w) && dec_sizes_ok((long long)B*16, h, w) && B < 65536 && h + 2 < 4*65536; }
size_t smd_conv3x3_thin_workspace_bytes(int B, int C, int h, int w) {
  if (!thin_sizes_ok(B, C, h, w)) return 0;
  return align256(smd::conv_thin_partials(B, C, h, w)*sizeof(float));
}
int smd_conv3x3_thin_bwd(const float* xp, const float* weight, const float* g_y, float* g_xp, float* g_weight, void* workspace, size_t workspace_bytes,
                         int B, int C, int h, int w, void* stream) {
  if (!g_y || (!g_xp && !g_weight) || (g_xp && !weight) || (g_weight && (!xp || !workspace))) return fail(SMD_E_INVALID, "null pointer");
  if (C != 16 && C != 32 && C >= 1) return fail(SMD_E_UNSUPPORTED, "the thin convolution serves 16 or 32 input channels, not %d", C);
  if (!thin_sizes_ok(B, C, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  if (g_weight && workspace_bytes < smd_conv3x3_thin_workspace_bytes(B, C, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (g_xp) { if (int rc = check_launch(smd::launch_conv_thin_bwd_data(g_y, weight, g_xp, B, C, h, w, st), "conv3x3_thin_bwd (data)")) return rc; }
  if (g_weight) { if (int rc = check_launch(smd::launch_conv_thin_bwd_wgt(xp, g_y, g_weight, (float*)workspace, B, C, h, w, st), "conv3x3_thin_bwd (weights)")) return rc; }
  return SMD_OK;
}
int smd_conv3x3_thin_fwd(const float* xp, const float* weight, float* y, int B, int C, int h, int w, void* stream) {
  if (!xp || !weight || !y) return fail(SMD_E_INVALID, "null pointer");
  if (C != 16 && C != 32 && C >= 1) return fail(SMD_E_UNSUPPORTED, "the thin convolution serves 16 or 32 input channels, not %d", C);
  if (!thin_sizes_ok(B, C, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d h=%d w=%d", B, C, h, w);
  return check_launch(smd::launch_conv_thin_fwd(xp, weight, y, B, C, h, w, (hipStream_t)stream), "conv3x3_thin_fwd");
}
static bool mfma_sizes_ok(int B, int C, int CO, int h, int w) {   // grid dimensions below 65536, element counts that int arithmetic inside a plane can hold,
  // one sample's activation and gradient under 2 GiB each (the weight gradient's LDS-DMA pieces carry 32-bit byte offsets from the sample's first element)
  return B >= 1 && C >= 1 && CO >= 1 && C <= 4096 && CO <= 4096 && h >= 1 && w >= 1 && h < 32768 && w < 32768 && (long long)(h + 2)*(w + 2) < (1ll << 30) &&
         (long long)B*((C + 31)/32)*((CO + 31)/32) < 65536 && (long long)C*(h + 2)*(w + 2) < (1ll << 29) && (long long)CO*h*w < (1ll << 29);
}
static bool mfma_fwd_served(int C, int CO) { return (C % 16 == 0 && CO % 32 == 0) || (CO == 16 && (C == 16 || C == 32)); }
static bool mfma_wgt_served(int C, int CO) { return CO % 32 == 0 || (CO == 16 && (C == 16 || C == 32)); }
static bool mfma_data_served(int C, int CO) { return (CO % 16 == 0 && C % 32 == 0) || (C == 16 && CO == 16); }
size_t smd_conv3x3_mfma_packed_bytes(int C, int CO, int pieces) {
  if (C < 1 || CO < 1 || pieces < 1 || pieces > 3) return 0;
  return align256(smd::conv_mfma_packed_elems(C, CO, pieces)*2);
}
size_t smd_conv3x3_mfma_workspace_bytes(int B, int C, int CO, int h, int w) {   // one size for the three operators
  if (!mfma_sizes_ok(B, C, CO, h, w)) return 0;
  smd::set_conv_two_tiles(knob("conv_two_tiles", 0));
  size_t n = 64;
  if (mfma_fwd_served(C, CO)) n = std::max(n, smd::conv_mfma_fwd_split_elems(B, C, CO, h, w));
  if (mfma_data_served(C, CO)) n = std::max(n, smd::conv_mfma_bwd_split_elems(B, C, CO, h, w));
  if (mfma_wgt_served(C, CO)) n = std::max(n, smd::conv_mfma_wgrad_partials(B, C, CO, h, w));
  return align256(n*sizeof(float));
}
int smd_conv3x3_mfma_pack(const float* weight, void* wp_fwd, void* wp_bwd, int C, int CO, int pieces, void* stream) {
  if (!weight || (!wp_fwd && !wp_bwd)) return fail(SMD_E_INVALID, "null pointer");
  if (pieces < 1 || pieces > 3) return fail(SMD_E_UNSUPPORTED, "pieces must be 1 (bfloat16 tensors), 2 or 3 (float tensors), not %d", pieces);
  if (C < 1 || CO < 1 || C > 4096 || CO > 4096) return fail(SMD_E_INVALID, "invalid sizes C=%d CO=%d", C, CO);
  if (wp_fwd && !mfma_fwd_served(C, CO)) return fail(SMD_E_UNSUPPORTED, "the forward form serves C %% 16 == 0 with CO %% 32 == 0, or CO == 16 with C == 16 | 32, not C=%d CO=%d", C, CO);
  if (wp_bwd && !mfma_data_served(C, CO)) return fail(SMD_E_UNSUPPORTED, "the data-gradient form serves CO %% 16 == 0 with C %% 32 == 0, or C == CO == 16, not C=%d CO=%d", C, CO);
  return check_launch(smd::launch_conv_mfma_pack(weight, wp_fwd, wp_bwd, C, CO, pieces, (hipStream_t)stream), "conv3x3_mfma_pack");
}
int smd_conv3x3_mfma_fwd(const void* xp, const void* wp_fwd, void* y, void* workspace, size_t workspace_bytes,
                         int B, int C, int CO, int h, int w, int pieces, void* stream) {
  if (!xp || !wp_fwd || !y || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (pieces < 1 || pieces > 3) return fail(SMD_E_UNSUPPORTED, "pieces must be 1 (bfloat16 tensors), 2 or 3 (float tensors), not %d", pieces);
  if (!mfma_sizes_ok(B, C, CO, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d CO=%d h=%d w=%d", B, C, CO, h, w);
  if (!mfma_fwd_served(C, CO)) return fail(SMD_E_UNSUPPORTED, "the forward serves C %% 16 == 0 with CO %% 32 == 0, or CO == 16 with C == 16 | 32, not C=%d CO=%d", C, CO);
  if (workspace_bytes < smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  smd::set_conv_two_tiles(knob("conv_two_tiles", 0));
  return check_launch(smd::launch_conv_mfma_fwd(xp, wp_fwd, y, (float*)workspace, B, C, CO, h, w, pieces, (hipStream_t)stream), "conv3x3_mfma_fwd");
}
int smd_conv3x3_mfma_bwd_data(const void* g_y, const void* wp_bwd, void* g_xp, void* workspace, size_t workspace_bytes,
                              int B, int C, int CO, int h, int w, int pieces, void* stream) {
  if (!g_y || !wp_bwd || !g_xp || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (pieces < 1 || pieces > 3) return fail(SMD_E_UNSUPPORTED, "pieces must be 1 (bfloat16 tensors), 2 or 3 (float tensors), not %d", pieces);
  if (!mfma_sizes_ok(B, C, CO, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d CO=%d h=%d w=%d", B, C, CO, h, w);
  if (!mfma_data_served(C, CO)) return fail(SMD_E_UNSUPPORTED, "the data gradient serves CO %% 16 == 0 with C %% 32 == 0, or C == CO == 16, not C=%d CO=%d", C, CO);
  if (workspace_bytes < smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  smd::set_conv_two_tiles(knob("conv_two_tiles", 0));
  return check_launch(smd::launch_conv_mfma_bwd_data(g_y, wp_bwd, g_xp, (float*)workspace, B, C, CO, h, w, pieces, (hipStream_t)stream), "conv3x3_mfma_bwd_data");
}
int smd_conv3x3_mfma_bwd_weight(const void* xp, const void* g_y, float* g_weight, void* workspace, size_t workspace_bytes,
                                int B, int C, int CO, int h, int w, int pieces, void* stream) {
  if (!xp || !g_y || !g_weight || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (pieces < 1 || pieces > 3) return fail(SMD_E_UNSUPPORTED, "pieces must be 1 (bfloat16 tensors), 2 or 3 (float tensors), not %d", pieces);
  if (!mfma_sizes_ok(B, C, CO, h, w)) return fail(SMD_E_INVALID, "invalid sizes B=%d C=%d CO=%d h=%d w=%d", B, C, CO, h, w);
  if (!mfma_wgt_served(C, CO)) return fail(SMD_E_UNSUPPORTED, "the weight gradient serves CO %% 32 == 0, or CO == 16 with C == 16 | 32, not C=%d CO=%d", C, CO);
  if (workspace_bytes < smd_conv3x3_mfma_workspace_bytes(B, C, CO, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_conv_mfma_bwd_wgt(xp, g_y, g_weight, (float*)workspace, B, C, CO, h, w, pieces, (hipStream_t)stream), "conv3x3_mfma_bwd_weight");
}
int smd_elu_up_cat_pad_fwd(const void* a, const float* bias, const void* skip, void* out, int B, int Ca, int Cs, int h, int w, int dtypes, void* stream) {
  if (!a || !out || (Cs > 0 && !skip)) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || Ca < 1 || Cs < 0 || h < 1 || w < 1 || !dec_sizes_ok((long long)B*(Ca + Cs), 2*h, 2*w))
    return fail(SMD_E_INVALID, "invalid sizes B=%d Ca=%d Cs=%d h=%d w=%d", B, Ca, Cs, h, w);
  return check_launch(smd::launch_elu_up_cat_pad_fwd(a, bias, skip, out, B, Ca, Cs, h, w, dtypes, (hipStream_t)stream), "elu_up_cat_pad_fwd");
}
int smd_elu_up_cat_pad_bwd(const void* a, const float* bias, const void* g_out, void* g_a, void* g_skip, float* g_bias,
                           void* workspace, size_t workspace_bytes, int B, int Ca, int Cs, int h, int w, int dtypes, void* stream) {
  if (!a || !g_out || (!g_a && !g_skip) || (g_bias && (!workspace || !g_a))) return fail(SMD_E_INVALID, "null pointer");
  if (B < 1 || Ca < 1 || Cs < 0 || h < 1 || w < 1 || !dec_sizes_ok((long long)B*(Ca + Cs), 2*h, 2*w))
    return fail(SMD_E_INVALID, "invalid sizes B=%d Ca=%d Cs=%d h=%d w=%d", B, Ca, Cs, h, w);
  if (g_bias && workspace_bytes < smd_decoder_glue_workspace_bytes(B, Ca, h, w)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_elu_up_cat_pad_bwd(a, bias, g_out, g_a, g_skip, g_bias, (float*)workspace, B, Ca, Cs, h, w, dtypes, (hipStream_t)stream),
                      "elu_up_cat_pad_bwd");
}

// ------------------------------------------------------------------------------------------------
// BatchNorm (+ residual, + ReLU)
size_t smd_bn_workspace_bytes(int N, int C, int HW) {
  if (N < 1 || C < 1 || HW < 1) return 0;
  return align256(((size_t)C*smd::bn_chunks(N, HW)*2 + (size_t)C*3)*sizeof(float));
}
int smd_bn_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
               int N, int C, int HW, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || C > 65535 || HW < 1 || (long long)N*HW < 2) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d HW=%d", N, C, HW);
  if (workspace_bytes < smd_bn_workspace_bytes(N, C, HW)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_bn_fwd(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, y, save_mean, save_invstd,
                                         (float*)workspace, N, C, HW, (hipStream_t)stream), "bn_fwd");
}
int smd_bn_bwd(const float* x, const float* y, const float* g_y, const float* gamma, const float* save_mean, const float* save_invstd, int relu,
               float* g_x, float* g_residual, float* g_gamma, float* g_beta, void* workspace, size_t workspace_bytes, int N, int C, int HW,
               void* stream) {
  if (!x || !g_y || !gamma || !save_mean || !save_invstd || !g_x || !g_gamma || !g_beta || !workspace || (relu && !y))
    return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || C > 65535 || HW < 1) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d HW=%d", N, C, HW);
  if (workspace_bytes < smd_bn_workspace_bytes(N, C, HW)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_bn_bwd(x, y, g_y, gamma, save_mean, save_invstd, relu, g_x, g_residual, g_gamma, g_beta, (float*)workspace,
                                         N, C, HW, (hipStream_t)stream), "bn_bwd");
}

int smd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int C, int H, int W, void* stream) {
  if (!x || !y || !idx) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || H < 1 || W < 1 || (long long)N*C*(((long long)H*W + 1023)/1024) >= (1ll << 31)) return fail(SMD_E_INVALID, "invalid sizes");
  return check_launch(smd::launch_maxpool_fwd(x, y, idx, (size_t)N*C, H, W, (hipStream_t)stream), "maxpool_fwd");
}
int smd_maxpool3x3s2_bwd(const float* g_y, const uint8_t* idx, float* g_x, int N, int C, int H, int W, void* stream) {
  if (!g_y || !idx || !g_x) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || H < 1 || W < 1 || (long long)N*C*(((long long)H*W + 1023)/1024) >= (1ll << 31)) return fail(SMD_E_INVALID, "invalid sizes");
  return check_launch(smd::launch_maxpool_bwd(g_y, idx, g_x, (size_t)N*C, H, W, (hipStream_t)stream), "maxpool_bwd");
}

// ------------------------------------------------------------------------------------------------
// Depthwise 7x7 convolution (ConvNeXt)
static bool dw_sizes_ok(int N, int C, int H, int W) {
  return N >= 1 && C >= 1 && H >= 1 && W >= 1 && (long long)N*C*smd::dwconv_tiles(H, W) < (1ll << 31) && (long long)H*W < (1ll << 30);
}
size_t smd_dwconv7x7_workspace_bytes(int C, int H, int W) {
  if (C < 1 || H < 1 || W < 1) return 0;
  return align256((size_t)C*smd::dwconv_tiles(H, W)*50*sizeof(float));
}
int smd_dwconv7x7_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int H, int W, int flip, void* stream) {
  if (!x || !weight || !y) return fail(SMD_E_INVALID, "null pointer");
  if (!dw_sizes_ok(N, C, H, W)) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d H=%d W=%d", N, C, H, W);
  return check_launch(smd::launch_dwconv7(x, weight, bias, y, N, C, H, W, flip, (hipStream_t)stream), "dwconv7x7_fwd");
}
int smd_dwconv7x7_wrw(const float* x, const float* g_y, float* g_weight, float* g_bias, void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, void* stream) {
  if (!x || !g_y || !g_weight || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (!dw_sizes_ok(N, C, H, W)) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d H=%d W=%d", N, C, H, W);
  if (workspace_bytes < smd_dwconv7x7_workspace_bytes(C, H, W)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_dwconv7_wrw(x, g_y, g_weight, g_bias, (float*)workspace, N, C, H, W, (hipStream_t)stream), "dwconv7x7_wrw");
}

// ------------------------------------------------------------------------------------------------
// Channel LayerNorm on NCHW (ConvNeXt)
size_t smd_layernorm_cf_workspace_bytes(int N, int C, int HW) {
  if (N < 1 || C < 1 || HW < 1) return 0;
  return align256((size_t)smd::ln_cf_chunks((size_t)N*HW)*C*2*sizeof(float));
}
int smd_layernorm_cf_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_is_bf16, float* mean, float* rstd, int N, int C, int HW,
                         float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || HW < 1 || (long long)N*HW >= (1ll << 38)) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d HW=%d", N, C, HW);
  return check_launch(smd::launch_ln_cf_fwd(x, gamma, beta, y, y_is_bf16, mean, rstd, N, C, HW, eps, (hipStream_t)stream), "layernorm_cf_fwd");
}
int smd_layernorm_cf_bwd(const float* x, const void* g_y, int g_y_is_bf16, const float* gamma, const float* mean, const float* rstd, float* g_x, float* g_gamma,
                         float* g_beta, void* workspace, size_t workspace_bytes, int N, int C, int HW, void* stream) {
  if (!x || !g_y || !gamma || !mean || !rstd || !g_x || !g_gamma || !g_beta || !workspace) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1 || C < 1 || C > 32767 || HW < 1) return fail(SMD_E_INVALID, "invalid sizes N=%d C=%d HW=%d", N, C, HW);
  if (workspace_bytes < smd_layernorm_cf_workspace_bytes(N, C, HW)) return fail(SMD_E_WORKSPACE, "workspace too small");
  return check_launch(smd::launch_ln_cf_bwd(x, g_y, g_y_is_bf16, gamma, mean, rstd, g_x, g_gamma, g_beta, (float*)workspace, N, C, HW, (hipStream_t)stream),
                      "layernorm_cf_bwd");
}

// ------------------------------------------------------------------------------------------------
// Pose / intrinsics prologue
int smd_pose_fwd(const float* aa, const float* t, const uint8_t* invert, int N, float* T, void* stream) {
  if (!aa || !t || !T) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1) return fail(SMD_E_INVALID, "invalid size N=%d", N);
  return check_launch(smd::launch_pose_fwd(aa, t, invert, N, T, (hipStream_t)stream), "pose_fwd");
}
int smd_pose_bwd(const float* aa, const float* t, const uint8_t* invert, int N, const float* g_T, float* g_aa, float* g_t, void* stream) {
  if (!aa || !t || !g_T || !g_aa || !g_t) return fail(SMD_E_INVALID, "null pointer");
  if (N < 1) return fail(SMD_E_INVALID, "invalid size N=%d", N);
  return check_launch(smd::launch_pose_bwd(aa, t, invert, N, g_T, g_aa, g_t, (hipStream_t)stream), "pose_bwd");
}
int smd_intrinsics_fwd(const float* fs, const float* cs, const float* K_in, int b, int h, int w, float* K, float* K_inv, void* stream) {
  if (!K_inv || (fs && (!cs || !K)) || (!fs && !K_in)) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1) return fail(SMD_E_INVALID, "invalid sizes");
  return check_launch(smd::launch_intrinsics_fwd(fs, cs, K_in, b, h, w, K, K_inv, (hipStream_t)stream), "intrinsics_fwd");
}
int smd_intrinsics_bwd(const float* fs, const float* cs, int b, int h, int w, const float* g_K, const float* g_Kinv,
                       float* g_fs, float* g_cs, void* stream) {
  if (!fs || !cs || !g_K || !g_Kinv || !g_fs || !g_cs) return fail(SMD_E_INVALID, "null pointer");
  if (b < 1 || h < 1 || w < 1) return fail(SMD_E_INVALID, "invalid sizes");
  return check_launch(smd::launch_intrinsics_bwd(fs, cs, b, h, w, g_K, g_Kinv, g_fs, g_cs, (hipStream_t)stream), "intrinsics_bwd");
}

// ------------------------------------------------------------------------------------------------
// Aspect-ratio augmentation
int smd_crop_resize(const float* const* src, float* const* dst, const int* planes, int nseg, int H, int W, int crop_h, int crop_w,
                    int out_h, int out_w, const float* K_in, float* K_out, int nK, void* stream) {
  if (!src || !dst || !planes) return fail(SMD_E_INVALID, "null pointer");
  if (nseg < 1 || nseg > smd::SMD_MAX_AR_SEGMENTS) return fail(SMD_E_INVALID, "nseg=%d outside [1, %d]", nseg, smd::SMD_MAX_AR_SEGMENTS);
  if (H < 2 || W < 2 || crop_h < 1 || crop_w < 1 || crop_h > H || crop_w > W || out_h < 1 || out_w < 1)
    return fail(SMD_E_INVALID, "invalid sizes: input %dx%d, crop %dx%d, output %dx%d", H, W, crop_h, crop_w, out_h, out_w);
  if ((K_in != nullptr) != (K_out != nullptr) || (K_in && nK < 1)) return fail(SMD_E_INVALID, "K_in / K_out / nK must be given together");
  smd::CropResizeArgs a;
  memset(&a, 0, sizeof(a));
  long long total = 0;
  for (int k = 0; k < nseg; ++k) {
    if (!src[k] || !dst[k] || planes[k] < 1) return fail(SMD_E_INVALID, "segment %d: null pointer or no planes", k);
    a.src[k] = src[k]; a.dst[k] = dst[k]; a.planes[k] = planes[k]; a.first_plane[k] = (int)total;
    total += planes[k];
  }
  if (total + 1 > 65535) return fail(SMD_E_INVALID, "too many image planes for one launch (%lld)", total);
  a.nseg = nseg; a.H = H; a.W = W; a.ch = crop_h; a.cw = crop_w; a.oh = out_h; a.ow = out_w;
  // kornia.geometry.transform.center_crop: start = int(src/2 - dst/2) (truncation); the window is then RESAMPLED (smd_aspect.hip), unless
  // it is the whole frame — the augmentation's resize-only branch (src/core/aspect_ratio.py:60)
  a.y0 = (int)((double)H/2.0 - (double)crop_h/2.0); a.x0 = (int)((double)W/2.0 - (double)crop_w/2.0);
  a.resample = (crop_h != H || crop_w != W) ? 1 : 0;
  a.K_in = K_in; a.K_out = K_out; a.nK = nK;
  return check_launch(smd::launch_crop_resize(a, (hipStream_t)stream), "crop_resize");
}

int smd_profile_enable(int which, int capacity) {
  if (which < 0 || which > 4 || capacity < 0) return fail(SMD_E_INVALID, "bad profile slot");
  ProfSlot& p = g_prof[which];
  for (int i = 0; i < 2*p.cap; ++i) (void)hipEventDestroy(p.ev[i]);
  delete[] p.ev;
  p = ProfSlot();
  if (capacity == 0) return SMD_OK;
  p.ev = new hipEvent_t[2*capacity];
  for (int i = 0; i < 2*capacity; ++i)
    if (hipEventCreate(&p.ev[i]) != hipSuccess) return fail(SMD_E_LAUNCH, "hipEventCreate failed");
  p.cap = capacity;
  return SMD_OK;
}

int smd_profile_collect(int which, float* ms_out, int max_out, int* n_out) {
  if (which < 0 || which > 4 || !ms_out || !n_out) return fail(SMD_E_INVALID, "bad profile arguments");
  ProfSlot& p = g_prof[which];
  int n = 0;
  for (int i = 0; i < p.used && n < max_out; ++i) {
    if (hipEventSynchronize(p.ev[2*i + 1]) != hipSuccess) return fail(SMD_E_LAUNCH, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[2*i], p.ev[2*i + 1]) != hipSuccess) return fail(SMD_E_LAUNCH, "hipEventElapsedTime failed");
    ms_out[n++] = ms;
  }
  p.used = 0;
  *n_out = n;
  return SMD_OK;
}

int smd_debug_stream_copy(const void* src, void* dst, size_t nbytes, int mode, void* stream) {
  if (!src || !dst) return fail(SMD_E_INVALID, "null pointer");
  if (nbytes < 16 || (nbytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return fail(SMD_E_INVALID, "size and pointers must be multiples of 16");
  return check_launch(smd::launch_stream_copy(src, dst, nbytes, mode, (hipStream_t)stream), "stream_copy");
}

int smd_debug_lane_shift(float* out_left, float* out_right, void* stream) {
  if (!out_left || !out_right) return fail(SMD_E_INVALID, "null pointer");
  return check_launch(smd::launch_debug_lane_shift(out_left, out_right, (hipStream_t)stream), "debug_lane_shift");
}

}  // extern "C"
