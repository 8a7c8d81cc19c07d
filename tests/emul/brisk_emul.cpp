// brisk_emul.cpp - TEST-ONLY host emulation of the HIP kernels' control flow.
//
// Runs the very same __host__ __device__ per-item functions the kernels use
// (ethzasl_brisk_amd/csrc/brisk_device_{detect,describe}.h) in plain loops that mirror the kernels
// in brisk_kernels.hip, so the order-faithful NMS logic, the float semantics and the pattern tables
// can be checked against the oracle in a container without a GPU.  It is never linked into or loaded
// by the product library; it is not a CPU fallback.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../ethzasl_brisk_amd/csrc/brisk_common.h"
#include "../../ethzasl_brisk_amd/csrc/brisk_device_describe.h"
#include "../../ethzasl_brisk_amd/csrc/brisk_device_detect.h"
#include "../../ethzasl_brisk_amd/csrc/brisk_pattern.h"

long brisk_cache_misses = 0;

namespace {

void make_geometry(int w, int h, int threshold, int octaves, BriskGeom* G) {  // == brisk_capi.hip
  memset(G, 0, sizeof(*G));
  G->nlayers = (octaves == 0) ? 1 : 2 * octaves;
  G->single_layer = (octaves == 0);
  G->w = w; G->h = h; G->threshold = threshold;
  G->lower_threshold = BRISK_LOWER_THRESHOLD;
  int off = 0;
  for (int l = 0; l < G->nlayers; ++l) {
    BriskLayerGeom& L = G->L[l];
    if (l == 0) { L.w = w; L.h = h; L.scale = 1.0f; L.offset = 0.0f; }
    else if (l == 1) {
      L.w = 2 * (G->L[0].w / 3); L.h = 2 * (G->L[0].h / 3);
      L.scale = (float)(G->L[0].scale * 1.5); L.offset = (float)(0.5 * L.scale - 0.5);
    } else {
      L.w = G->L[l - 2].w / 2; L.h = G->L[l - 2].h / 2;
      L.scale = G->L[l - 2].scale * 2; L.offset = (float)(0.5 * L.scale - 0.5);
    }
    L.stride = brisk_align_up(L.w > 0 ? L.w : 1, BRISK_STRIDE_ALIGN);
    L.off = off;
    off += brisk_align_up(L.stride * (L.h > 0 ? L.h : 1), 256);
  }
  G->pyr_elems = off;
}

struct Emul {
  BriskGeom G;
  std::vector<uint8_t> pyr;
  std::vector<uint16_t> smap;
  std::vector<BriskCand> cand;
  std::vector<int> ties[BRISK_MAX_LAYERS];
  std::vector<BriskKeyPoint> kps;
  int relax_iters = 0, max_chain = 0;
  bool undefined = false;  // ordered path: the reference has no defined result on this input
};

BriskLayerView view(Emul& E, int l) {
  BriskLayerView v;
  v.img = E.pyr.data() + E.G.L[l].off;
  v.smap = E.smap.data() + E.G.L[l].off;
  v.w = E.G.L[l].w; v.h = E.G.L[l].h; v.stride = E.G.L[l].stride;
  brisk_block_clear(&v.blk); brisk_block_clear(&v.blk58); v.miss = 0;
  return v;
}

void touch_apply(Emul& E, int l_above, int x0, int y0, unsigned mask) {
  BriskLayerView La = view(E, l_above);
  for (int b = 0; b < 16; ++b)
    if (mask & (1u << b)) La.smap[(long)(y0 + (b >> 2)) * La.stride + x0 + (b & 3)] |= BRISK_SM_TOUCH;
}

// mirrors k_copy_layer0 + k_pyramid_level + k_detect + k_classify_refine + k_tie_resolve + k_finalize
void run_detect(Emul& E, const uint8_t* img, int w, int h, int threshold, int octaves, unsigned shuffle_seed, int jacobi,
                bool use_cache, bool ordered = false, bool jacobi_no_scale_nms = false, int lower_threshold = BRISK_LOWER_THRESHOLD) {
  make_geometry(w, h, threshold, octaves, &E.G);
  E.G.lower_threshold = lower_threshold;
  const BriskGeom& G = E.G;
  E.pyr.assign((size_t)G.pyr_elems + 256, 0);
  E.smap.assign((size_t)G.pyr_elems + 256, 0);
  for (int y = 0; y < h; ++y) memcpy(E.pyr.data() + G.L[0].off + (size_t)y * G.L[0].stride, img + (size_t)y * w, w);
  for (int l = 1; l < G.nlayers; ++l) {
    const int sl = (l == 1) ? 0 : l - 2;
    const uint8_t* src = E.pyr.data() + G.L[sl].off;
    uint8_t* dst = E.pyr.data() + G.L[l].off;
    for (int y = 0; y < G.L[l].h; ++y)
      for (int x = 0; x < G.L[l].w; ++x)
        dst[(size_t)y * G.L[l].stride + x] = (l == 1) ? brisk_twothird_px(src, G.L[sl].stride, G.L[sl].w, x, y)
                                                      : brisk_half_px(src, G.L[sl].stride, G.L[sl].w, x, y);
  }
  // k_detect
  E.cand.clear();
  for (int l = 0; l < G.nlayers; ++l) {
    BriskLayerView L = view(E, l);
    for (int y = 3; y <= L.h - 4; ++y)
      for (int x = 3; x <= L.w - 4; ++x) {
        const int D = brisk_detect_px(L.img + (long)y * L.stride + x, L.stride, threshold, E.G.lower_threshold);
        if (!D) continue;
        L.smap[(long)y * L.stride + x] = (uint16_t)D;
        BriskCand c;
        memset(&c, 0, sizeof(c));
        c.x = (uint16_t)x; c.y = (uint16_t)y; c.layer = (uint8_t)l; c.D = (uint8_t)D;
        c.key = ((unsigned)l << 26) | ((unsigned)y << 13) | (unsigned)x;
        E.cand.push_back(c);
      }
  }
  if (shuffle_seed) {  // atomic-append order is arbitrary on the GPU
    srand(shuffle_seed);
    for (size_t i = E.cand.size(); i > 1; --i) std::swap(E.cand[i - 1], E.cand[(size_t)rand() % i]);
  }
  if (ordered) {  // k_order_candidates + k_ordered_keypoints: the sequential algorithm on the literal cache
    std::sort(E.cand.begin(), E.cand.end(), [](const BriskCand& a, const BriskCand& b) { return a.key < b.key; });
    std::vector<unsigned> order(E.cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (unsigned)i;
    E.kps.assign(E.cand.size() + 1, BriskKeyPoint());
    BriskOrderedOut out;
    out.kp = E.kps.data(); out.cap = (int)E.kps.size(); out.n = 0; out.mask = nullptr; out.mask_row_pitch = 0;
    E.G.no_scale_nms = (jacobi_no_scale_nms && octaves != 0) ? 1 : 0;
    E.undefined = brisk_ordered_walk(E.G, E.pyr.data(), E.smap.data(), E.cand.data(), order.data(), (int)E.cand.size(),
                                     E.G.no_scale_nms != 0, &out);
    E.kps.resize((size_t)out.n);
    return;
  }
  // k_classify_refine (one wave per candidate: lane-parallel score blocks, then uniform scalar logic)
  for (int l = 0; l < G.nlayers; ++l) E.ties[l].clear();
  for (size_t i = 0; i < E.cand.size(); ++i) {
    BriskCand* c = &E.cand[i];
    const int l = c->layer, x = c->x, y = c->y, D = c->D;
    const bool has_above = !G.single_layer && (l + 1 < G.nlayers);
    const bool has_below = !G.single_layer && (l > 0);
    BriskLayerView Lo = view(E, l), La = view(E, has_above ? l + 1 : l), Lb = view(E, has_below ? l - 1 : l);
    uint8_t c_own[9], c_58[9], c_above[16], c_below[16];
    int aax = 0, aay = 0, bax = 0, bay = 0;
    if (has_above) brisk_block_anchor(true, (l & 1) != 0, x, y, &aax, &aay);
    if (has_below) brisk_block_anchor(false, (l & 1) != 0, x, y, &bax, &bay);
    if (use_cache) {
      for (int k = 0; k < 9; ++k) c_own[k] = (uint8_t)brisk_V_eval(Lo, x - 1 + k % 3, y - 1 + k / 3);
      if (l == 0 && !G.single_layer) for (int k = 0; k < 9; ++k) c_58[k] = (uint8_t)brisk_V58_eval(Lo, x - 1 + k % 3, y - 1 + k / 3);
      if (has_above) for (int k = 0; k < 16; ++k) c_above[k] = (uint8_t)brisk_V_eval(La, aax + (k & 3), aay + (k >> 2));
      if (has_below) for (int k = 0; k < 16; ++k) c_below[k] = (uint8_t)brisk_V_eval(Lb, bax + (k & 3), bay + (k >> 2));
      brisk_block_from_bytes(&Lo.blk, c_own, 9, x - 1, y - 1, 3, 3);
      if (l == 0 && !G.single_layer) brisk_block_from_bytes(&Lo.blk58, c_58, 9, x - 1, y - 1, 3, 3);
      if (has_above) brisk_block_from_bytes(&La.blk, c_above, 16, aax, aay, 4, 4);
      if (has_below) brisk_block_from_bytes(&Lb.blk, c_below, 16, bax, bay, 4, 4);
    }
    int nprobed;
    const unsigned status = use_cache ? brisk_classify<false>(Lo, x, y, D, &nprobed) : brisk_classify<true>(Lo, x, y, D, &nprobed);
    unsigned bits = ((unsigned)nprobed << 8) | (status << 12);
    unsigned flags = 0;
    if (status != BRISK_ST_REJ) {
      BriskKeyPoint kp;
      bool e5;
      BriskTouch touch;
      touch.on = false; touch.mask = 0; touch.x0 = 0; touch.y0 = 0;
      const bool ok = use_cache ? brisk_refine<false>(G, Lb, Lo, La, l, x, y, &kp, &e5, &touch)
                                : brisk_refine<true>(G, Lb, Lo, La, l, x, y, &kp, &e5, &touch);
      brisk_cache_misses += Lo.miss + La.miss + Lb.miss;
      if (ok) { flags |= 1; c->kx = kp.x; c->ky = kp.y; c->ksize = kp.size; c->kresp = kp.response; }
      if (e5) { flags |= 2; bits |= BRISK_SM_E5; }
      c->fp_x0 = (int16_t)touch.x0; c->fp_y0 = (int16_t)touch.y0; c->fp_mask = (uint16_t)touch.mask;
      if (status == BRISK_ST_PASS && touch.mask) touch_apply(E, l + 1, touch.x0, touch.y0, touch.mask);
      if (status == BRISK_ST_TIE) E.ties[l].push_back((int)i);
    }
    c->status = (uint8_t)status;
    c->flags = (uint8_t)flags;
    view(E, l).smap[(long)y * G.L[l].stride + x] |= (uint16_t)bits;
  }
  BriskLayerView Lv[BRISK_MAX_LAYERS];
  for (int l = 0; l < G.nlayers; ++l) Lv[l] = view(E, l);
  // k_tie_resolve
  E.relax_iters = 0;
  for (int l = 0; l < G.nlayers; ++l) {
    const BriskLayerView L = Lv[l];
    const bool last = (l == G.nlayers - 1);
    const bool float_patch = last || G.single_layer;
    const bool touch2x2 = last && !G.single_layer;
    if (jacobi == 2) {  // kernel's main path: raster-sorted, in order, lane-split evaluation
      std::vector<std::pair<unsigned, int>> order;
      for (int idx : E.ties[l]) order.push_back({E.cand[idx].key, idx});
      std::sort(order.begin(), order.end());
      for (auto& o : order) {
        BriskCand* c = &E.cand[o.second];
        const int cx = c->x, cy = c->y;
        uint16_t wl[81];
        for (int dy = -4; dy <= 4; ++dy)
          for (int dx = -4; dx <= 4; ++dx) {
            const int qx = cx + dx, qy = cy + dy;
            unsigned v = 0;
            if (qx >= 0 && qy >= 0 && qx < L.w && qy < L.h) v = L.smap[(long)qy * L.stride + qx];
            wl[(dy + 4) * 9 + dx + 4] = (uint16_t)v;
            if ((dy < 0 || (dy == 0 && dx < 0)) && BRISK_SM_D(v) && BRISK_SM_STATUS(v) == BRISK_ST_TIE) abort();  // would spin forever
          }
        const int centre = c->D;
        int sv[8], raw[25];
        uint8_t kp5[25];  // k_score_blocks bytes 0-24
        for (int q = 0; q < 25; ++q) kp5[q] = (uint8_t)brisk_V_eval(L, cx - 2 + q % 5, cy - 2 + q / 5);
        for (int k = 0; k < 8; ++k) sv[k] = brisk_tie_probe_value<false>(L, float_patch, touch2x2, cx, cy, centre, k, wl, cx - 4, cy - 4, 9, kp5);
        for (int q = 0; q < 25; ++q) raw[q] = brisk_tie_raw_value<false>(L, float_patch, touch2x2, cx, cy, centre, q, wl, cx - 4, cy - 4, 9, kp5);
        if (brisk_tie_decide(centre, sv, raw)) {
          if (c->fp_mask && l + 1 < G.nlayers) touch_apply(E, l + 1, c->fp_x0, c->fp_y0, c->fp_mask);
          c->status = BRISK_ST_PASS;
          L.smap[(long)cy * L.stride + cx] ^= 0x3000u;
        } else {
          c->status = BRISK_ST_FAIL;
          L.smap[(long)cy * L.stride + cx] |= 0x1000u;
        }
      }
      continue;
    }
    for (int iter = 0; iter < 100000; ++iter) {
      int remaining = 0, progressed = 0;
      std::vector<std::pair<int, bool>> decided;  // jacobi mode: apply after the sweep
      for (int idx : E.ties[l]) {
        BriskCand* c = &E.cand[idx];
        if (c->status != BRISK_ST_TIE) continue;
        const int cx = c->x, cy = c->y;
        uint16_t wl[81];
        bool ready = true;
        for (int dy = -4; dy <= 4; ++dy)
          for (int dx = -4; dx <= 4; ++dx) {
            const int qx = cx + dx, qy = cy + dy;
            unsigned v = 0;
            if (qx >= 0 && qy >= 0 && qx < L.w && qy < L.h) v = L.smap[(long)qy * L.stride + qx];
            wl[(dy + 4) * 9 + dx + 4] = (uint16_t)v;
            if ((dy < 0 || (dy == 0 && dx < 0)) && BRISK_SM_D(v) && BRISK_SM_STATUS(v) == BRISK_ST_TIE) ready = false;
          }
        if (!ready) { remaining++; continue; }
        const bool pass = brisk_tie_eval<true>(L, float_patch, touch2x2, cx, cy, wl, cx - 4, cy - 4, 9, nullptr);
        progressed++;
        if (jacobi) { decided.push_back({idx, pass}); continue; }
        if (pass) {
          if (c->fp_mask && l + 1 < G.nlayers) touch_apply(E, l + 1, c->fp_x0, c->fp_y0, c->fp_mask);
          c->status = BRISK_ST_PASS;
          L.smap[(long)cy * L.stride + cx] ^= 0x3000u;
        } else {
          c->status = BRISK_ST_FAIL;
          L.smap[(long)cy * L.stride + cx] |= 0x1000u;
        }
      }
      for (auto& d : decided) {
        BriskCand* c = &E.cand[d.first];
        if (d.second) {
          if (c->fp_mask && l + 1 < G.nlayers) touch_apply(E, l + 1, c->fp_x0, c->fp_y0, c->fp_mask);
          c->status = BRISK_ST_PASS;
          L.smap[(long)c->y * L.stride + c->x] ^= 0x3000u;
        } else {
          c->status = BRISK_ST_FAIL;
          L.smap[(long)c->y * L.stride + c->x] |= 0x1000u;
        }
      }
      if (progressed) E.relax_iters++;
      if (iter + 1 > E.max_chain && progressed) E.max_chain = iter + 1;
      if (remaining == 0 || progressed == 0) break;
    }
  }
  // k_finalize
  std::vector<std::pair<unsigned, int>> keys;
  for (size_t i = 0; i < E.cand.size(); ++i)
    if (E.cand[i].status == BRISK_ST_PASS && (E.cand[i].flags & 1)) keys.push_back({E.cand[i].key, (int)i});
  std::sort(keys.begin(), keys.end());
  E.kps.clear();
  for (auto& k : keys) {
    const BriskCand& c = E.cand[k.second];
    BriskKeyPoint kp;
    kp.x = c.kx; kp.y = c.ky; kp.size = c.ksize; kp.angle = -1.0f; kp.response = c.kresp;
    kp.octave = G.single_layer ? 0 : c.layer; kp.class_id = -1;
    E.kps.push_back(kp);
  }
}

struct EmulPattern {
  BriskPatternHost H;
  BriskPatternDev P;
};

}  // namespace

extern "C" {

// returns keypoint count; *out malloc'd (free with emul_free). stats[0]=#candidates, [1]=#ties, [2]=relaxation sweeps,
// [3]=longest per-layer chain, [4]=score-block cache misses, [5]=candidates that stored a score <= 2 (BriskFrameCounters::low_score
// on the device: such a frame takes the ordered path).  mode: bits 0-1 tie scheme (0 Gauss-Seidel sweeps, 1 Jacobi
// sweeps, 2 sorted in-order = the kernel's main path), bit 2 = use the lane-parallel score-block caches, bit 3 = the
// ordered path of thresholds below 20 (k_order_candidates + k_ordered_keypoints), bit 4 = the ordered path's
// suppressScaleNonmaxima = false branch (returns -1 where the reference has no defined result)
int emul_detect(const uint8_t* img, int w, int h, int threshold, int octaves, unsigned shuffle_seed, int jacobi,
                BriskKeyPoint** out, int* stats) {
  Emul E;
  brisk_cache_misses = 0;
  run_detect(E, img, w, h, threshold, octaves, shuffle_seed, jacobi & 3, (jacobi & 4) != 0, (jacobi & 24) != 0, (jacobi & 16) != 0);
  if (E.undefined) return -1;
  *out = (BriskKeyPoint*)malloc(sizeof(BriskKeyPoint) * (E.kps.size() + 1));
  memcpy(*out, E.kps.data(), sizeof(BriskKeyPoint) * E.kps.size());
  if (stats) {
    stats[0] = (int)E.cand.size();
    int nt = 0;
    for (int l = 0; l < E.G.nlayers; ++l) nt += (int)E.ties[l].size();
    stats[1] = nt; stats[2] = E.relax_iters; stats[3] = E.max_chain; stats[4] = (int)brisk_cache_misses;
    int low = 0;
    for (const BriskCand& c : E.cand) low += (c.D <= 2) ? 1 : 0;
    stats[5] = low;
  }
  return (int)E.kps.size();
}

// k_tie_resolve lets a tie of layer l + 1 in row cy start as soon as the layer below has finished its rows
// < brisk_tie_rows_needed(cy): every candidate of layer l whose touch footprint (4x4 block on layer l + 1, recorded by
// the refinement code itself) has a pixel within 2 rows of cy must lie in such a row.  Returns the violations found
// (a) on the candidates of an image, (b) on the footprint formula for every row up to the engine's size limit.
int emul_tie_rows_needed_violations(const uint8_t* img, int w, int h, int threshold, int octaves) {
  Emul E;
  brisk_cache_misses = 0;
  run_detect(E, img, w, h, threshold, octaves, 0, 2, true, false, false);
  int bad = 0;
  for (const BriskCand& c : E.cand) {
    if (!c.fp_mask || c.layer + 1 >= E.G.nlayers) continue;
    const int ha = E.G.L[c.layer + 1].h;
    for (int b = 0; b < 16; ++b) {
      if (!(c.fp_mask & (1u << b))) continue;
      const int row = c.fp_y0 + (b >> 2);
      for (int cy = row - 2; cy <= row + 2; ++cy)
        if (cy >= 0 && cy < ha && !((int)c.y < brisk_tie_rows_needed(cy, (c.layer & 1) == 0))) ++bad;
    }
  }
  for (int y = 0; y < 8192; ++y)
    for (int oct = 0; oct < 2; ++oct) {
      // first row of the footprint (brisk_score_max_other, `above` branch)
      const float y_1 = oct ? (float)((float)(4 * y - 1 - 2) / 6.0) : (float)(6 * y - 1 - 3) / 8.0f;
      const int y0 = (int)y_1 - 1;
      for (int row = y0; row <= y0 + 3; ++row)
        for (int cy = row - 2; cy <= row + 2; ++cy)
          if (cy >= 0 && !(y < brisk_tie_rows_needed(cy, oct != 0))) ++bad;
    }
  return bad;
}

// mirrors brisk_hip_compute_scale: pyramid with lower threshold 0, then the ComputeScale walk on one "lane";
// returns -1 where the reference has no defined result
int emul_compute_scale(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress, const BriskKeyPoint* in,
                       int n_in, BriskKeyPoint** out) {
  Emul E;
  if (n_in == 0) {  // empty list: plain detection on the lower-threshold-0 pyramid, ordered path
    run_detect(E, img, w, h, threshold, octaves, 0, 0, false, true, !suppress, 0);
    if (E.undefined) return -1;
  } else {
    make_geometry(w, h, threshold, octaves, &E.G);
    E.G.lower_threshold = 0;
    const BriskGeom& G = E.G;
    E.pyr.assign((size_t)G.pyr_elems + 256, 0);
    E.smap.assign((size_t)G.pyr_elems + 256, 0);
    for (int y = 0; y < h; ++y) memcpy(E.pyr.data() + G.L[0].off + (size_t)y * G.L[0].stride, img + (size_t)y * w, w);
    for (int l = 1; l < G.nlayers; ++l) {
      const int sl = (l == 1) ? 0 : l - 2;
      const uint8_t* src = E.pyr.data() + G.L[sl].off;
      uint8_t* dst = E.pyr.data() + G.L[l].off;
      for (int y = 0; y < G.L[l].h; ++y)
        for (int x = 0; x < G.L[l].w; ++x)
          dst[(size_t)y * G.L[l].stride + x] = (l == 1) ? brisk_twothird_px(src, G.L[sl].stride, G.L[sl].w, x, y)
                                                        : brisk_half_px(src, G.L[sl].stride, G.L[sl].w, x, y);
    }
    std::vector<uint32_t> det(1 << 20);
    E.kps.assign((size_t)n_in * G.nlayers + det.size() + 1, BriskKeyPoint());
    BriskOrderedOut o;
    o.kp = E.kps.data(); o.cap = (int)E.kps.size(); o.n = 0; o.mask = nullptr; o.mask_row_pitch = 0;
    bool cap_exceeded = false;
    if (brisk_compute_scale_walk(G, E.pyr.data(), E.smap.data(), in, n_in, suppress != 0, det.data(), (int)det.size(), &o,
                                 &cap_exceeded) || cap_exceeded)
      return -1;
    E.kps.resize((size_t)o.n);
  }
  *out = (BriskKeyPoint*)malloc(sizeof(BriskKeyPoint) * (E.kps.size() + 1));
  memcpy(*out, E.kps.data(), sizeof(BriskKeyPoint) * E.kps.size());
  return (int)E.kps.size();
}

// The same call in the pieces the k_cs_* kernels run one lane per (layer, provided point), with the items of every phase
// taken in a pseudo-random order (seed): the claim behind those kernels is that the phases are order-free among themselves.
// Returns -1 where the reference has no defined result, -2 where a layer admits no provided point (the engine then runs the
// one-lane walk instead).
int emul_compute_scale_phased(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress, const BriskKeyPoint* in,
                              int n_in, unsigned seed, BriskKeyPoint** out) {
  Emul E;
  make_geometry(w, h, threshold, octaves, &E.G);
  E.G.lower_threshold = 0;
  const BriskGeom& G = E.G;
  E.pyr.assign((size_t)G.pyr_elems + 256, 0);
  E.smap.assign((size_t)G.pyr_elems + 256, 0);
  for (int y = 0; y < h; ++y) memcpy(E.pyr.data() + G.L[0].off + (size_t)y * G.L[0].stride, img + (size_t)y * w, w);
  for (int l = 1; l < G.nlayers; ++l) {
    const int sl = (l == 1) ? 0 : l - 2;
    const uint8_t* src = E.pyr.data() + G.L[sl].off;
    uint8_t* dst = E.pyr.data() + G.L[l].off;
    for (int y = 0; y < G.L[l].h; ++y)
      for (int x = 0; x < G.L[l].w; ++x)
        dst[(size_t)y * G.L[l].stride + x] = (l == 1) ? brisk_twothird_px(src, G.L[sl].stride, G.L[sl].w, x, y)
                                                      : brisk_half_px(src, G.L[sl].stride, G.L[sl].w, x, y);
  }
  struct Item { int l, j, k; };
  std::vector<Item> items;
  std::vector<std::vector<int> > adm((size_t)G.nlayers);
  for (int l = 0; l < G.nlayers; ++l) {
    for (int k = 0; k < n_in; ++k) {
      float kx, ky;
      if (brisk_provided_on_layer(G, l, in[k], &kx, &ky)) { items.push_back({l, (int)adm[(size_t)l].size(), k}); adm[(size_t)l].push_back(k); }
    }
    if (adm[(size_t)l].empty()) return -2;
  }
  unsigned st = seed * 2654435761u + 99u;
  auto shuffle = [&]() {
    for (size_t i = items.size(); i > 1; --i) {
      st = st * 1664525u + 1013904223u;
      std::swap(items[i - 1], items[(st >> 8) % i]);
    }
  };
  shuffle();
  for (const Item& it : items) {
    float kx, ky;
    brisk_provided_on_layer(G, it.l, in[it.k], &kx, &ky);
    brisk_cs_touch(G, E.pyr.data(), E.smap.data(), it.l, kx, ky);
  }
  shuffle();
  for (const Item& it : items) {
    float kx, ky;
    brisk_provided_on_layer(G, it.l, in[it.k], &kx, &ky);
    if (brisk_cs_score(G, E.pyr.data(), E.smap.data(), it.l, kx, ky)) return -1;
  }
  const bool flat = !suppress && !G.single_layer;
  std::vector<std::vector<BriskKeyPoint> > res((size_t)G.nlayers);
  std::vector<std::vector<char> > ok((size_t)G.nlayers);
  for (int l = 0; l < G.nlayers; ++l) {
    if (flat && adm[(size_t)l].size() > adm[0].size()) return -1;
    res[(size_t)l].resize(adm[(size_t)l].size());
    ok[(size_t)l].assign(adm[(size_t)l].size(), 0);
  }
  shuffle();
  for (const Item& it : items) {
    float kx, ky;
    BriskKeyPoint kp;
    if (flat) {
      const BriskKeyPoint& src0 = in[adm[0][(size_t)it.j]];
      brisk_provided_on_layer(G, 0, src0, &kx, &ky);
      brisk_cs_flat(G, E.pyr.data(), E.smap.data(), it.l, src0, kx, ky, &kp);
      ok[(size_t)it.l][(size_t)it.j] = 1;
    } else {
      brisk_provided_on_layer(G, it.l, in[it.k], &kx, &ky);
      ok[(size_t)it.l][(size_t)it.j] = brisk_cs_refine(G, E.pyr.data(), E.smap.data(), it.l, in[it.k], kx, ky, &kp) ? 1 : 0;
    }
    res[(size_t)it.l][(size_t)it.j] = kp;
  }
  std::vector<BriskKeyPoint> all;
  for (int l = 0; l < G.nlayers; ++l)
    for (size_t j = 0; j < res[(size_t)l].size(); ++j)
      if (ok[(size_t)l][j]) all.push_back(res[(size_t)l][j]);
  *out = (BriskKeyPoint*)malloc(sizeof(BriskKeyPoint) * (all.size() + 1));
  memcpy(*out, all.data(), sizeof(BriskKeyPoint) * all.size());
  return (int)all.size();
}

void emul_free(void* p) { free(p); }

// the kernel evaluates the tie list on 8 lanes (brisk_tie_neighbour_ok); must equal the serial brisk_tie_decide
int emul_tie_decide_mismatches(unsigned seed, int n) {
  int bad = 0;
  unsigned st = seed * 2654435761u + 12345u;
  for (int it = 0; it < n; ++it) {
    int s[8], raw[25];
    st = st * 1664525u + 1013904223u;
    const int centre = 3 + (int)((st >> 8) % 60);
    for (int k = 0; k < 8; ++k) { st = st * 1664525u + 1013904223u; const unsigned r = (st >> 10) % 4; s[k] = r == 0 ? centre : r == 1 ? 0 : (int)((st >> 16) % (unsigned)(centre + 1)); }
    for (int q = 0; q < 25; ++q) { st = st * 1664525u + 1013904223u; raw[q] = ((st >> 9) & 1) ? 0 : (int)((st >> 12) % 80); }
    bool all_ok = true;
    for (int o = 0; o < 8; ++o) all_ok = all_ok && brisk_tie_neighbour_ok(centre, s, raw, o);
    bad += (all_ok != brisk_tie_decide(centre, s, raw));
  }
  return bad;
}

// exhaustive check of the multiply-free threshold scaling used by k_detect; returns the number of mismatches
int emul_b2_fast_mismatches(void) {
  int bad = 0;
  for (int thr = 1; thr <= 255; ++thr) {
    const float k = brisk_b2_factor(thr);
    for (int tc = 0; tc <= 255; ++tc) bad += (brisk_b2_fast(tc, k) != (tc * thr) / 100);
  }
  return bad;
}

// The tie kernel's two-step replay (brisk_state_static before the wait for pending ties, brisk_state_resolve after it)
// against the one-step brisk_state_at, on random 9x9 windows: `n` random candidates, all 33 slots, the three touch
// geometries; raster-earlier decided ties are shown as pending (status TIE) to the static step.  Returns mismatches.
int emul_state_split_mismatches(unsigned seed, int n) {
  auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  int bad = 0;
  for (int it = 0; it < n; ++it) {
    BriskLayerView L;
    memset(&L, 0, sizeof(L));
    L.w = 12 + (int)(rnd() % 30);
    L.h = 12 + (int)(rnd() % 30);
    L.stride = L.w;
    const int cx = 3 + (int)(rnd() % (unsigned)(L.w - 6)), cy = 3 + (int)(rnd() % (unsigned)(L.h - 6));
    uint16_t wfin[81], wpre[81];
    const unsigned density = 2 + rnd() % 6;
    for (int e = 0; e < 81; ++e) {
      const int dy = e / 9 - 4, dx = e % 9 - 4;
      const int qx = cx + dx, qy = cy + dy;
      unsigned v = 0;
      const bool inside = qx >= 0 && qy >= 0 && qx < L.w && qy < L.h;
      const bool centre = (dx == 0 && dy == 0);
      const bool earlier = dy < 0 || (dy == 0 && dx < 0);
      if (inside && (centre || rnd() % 8 < density)) {
        const unsigned dsel = rnd() % 4;
        const unsigned D = dsel == 0 ? 1 + rnd() % 2 : dsel == 1 ? 3 + rnd() % 3 : 1 + rnd() % 255;
        const unsigned st = centre ? BRISK_ST_TIE : earlier ? (unsigned[]){BRISK_ST_REJ, BRISK_ST_PASS, BRISK_ST_FAIL, BRISK_ST_PASS}[rnd() % 4] : rnd() % 4;
        v = D | ((rnd() % 9) << 8) | (st << 12) | ((rnd() % 2) ? BRISK_SM_E5 : 0u);
      }
      if (inside && rnd() % 4 == 0) v |= BRISK_SM_TOUCH;
      wfin[e] = (uint16_t)v;
      unsigned vp = v;
      const unsigned st = BRISK_SM_STATUS(v);
      if (earlier && BRISK_SM_D(v) && (st == BRISK_ST_PASS || st == BRISK_ST_FAIL) && rnd() % 2) vp = (v & ~0x3000u) | (BRISK_ST_TIE << 12);
      wpre[e] = (uint16_t)vp;
    }
    uint8_t kp5[25];
    for (int q = 0; q < 25; ++q) kp5[q] = (uint8_t)((rnd() % 3 == 0) ? rnd() % 6 : rnd() % 256);
    const int centre = BRISK_SM_D(wfin[40]);
    for (int geo = 0; geo < 3; ++geo) {
      const bool float_patch = geo > 0, touch2x2 = geo == 2;
      for (int slot = 0; slot < 33; ++slot) {
        const int ref = brisk_tie_slot_value<false>(L, float_patch, touch2x2, cx, cy, centre, slot, wfin, cx - 4, cy - 4, 9, kp5);
        unsigned dyn = 0;
        const unsigned pk = brisk_tie_slot_static(L, float_patch, touch2x2, cx, cy, slot, wpre, cx - 4, cy - 4, 9, kp5, &dyn);
        const int got = brisk_tie_slot_resolve(L, pk, dyn, cx, cy, centre, slot, wfin, cx - 4, cy - 4, 9, kp5);
        if (ref != got) ++bad;
        // the kernel's form: the slot's act / not-self masks precomputed (brisk_state_masks)
        int sx, sy; bool own; unsigned am, ns, dyn2 = 0;
        brisk_tie_slot_offset(slot, &sx, &sy, &own);
        brisk_state_masks(sx, sy, own, &am, &ns);
        const unsigned pk2 = brisk_tie_slot_static_m(L, float_patch, touch2x2, cx, cy, slot, sx, sy, am, ns, wpre, cx - 4, cy - 4, 9, kp5, &dyn2);
        if (pk2 != pk || dyn2 != dyn) ++bad;
      }
    }
  }
  return bad;
}

// The block form of GetScoreMaxAbove (brisk_score_max_above_blk, what k_classify_refine runs) against the generic
// brisk_score_max_other<0>(above = true) on random 4 x 4 score blocks: candidates anywhere on random layers (incl. next to
// the borders, where touches are not recorded and the block holds zeros), both layer parities, centre scores around the
// block's values so that the abort rule fires at every grid position.  Everything the caller sees is compared: the value,
// ismax, the offsets (bit patterns), the touch record and the miss flag.  Returns mismatches.
int emul_score_max_above_blk_mismatches(unsigned seed, int n, int* stats /* 4: maxima, aborted scans, misses, touch masks that differ from 0 */) {
  auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  int bad = 0;
  for (int it = 0; it < n; ++it) {
    const bool odd = rnd() & 1;
    // the layer above: w x h; the candidate (x, y) on the layer below it, anywhere the detector can place one
    BriskLayerView La;
    memset(&La, 0, sizeof(La));
    La.w = 8 + (int)(rnd() % 60);
    La.h = 8 + (int)(rnd() % 60);
    La.stride = La.w;
    // own layer is 1.5 (even -> odd above) or 4/3 (odd -> even above) times as large; keep the window inside [0, w)
    const int ow = odd ? (La.w * 4) / 3 : (La.w * 3) / 2, oh = odd ? (La.h * 4) / 3 : (La.h * 3) / 2;
    const int x = 3 + (int)(rnd() % (unsigned)brisk_max(ow - 6, 1)), y = 3 + (int)(rnd() % (unsigned)brisk_max(oh - 6, 1));
    int ax, ay;
    brisk_block_anchor(true, odd, x, y, &ax, &ay);
    uint8_t vals[16];
    const unsigned mode = rnd() % 4;
    const int base = (int)(rnd() % 200);
    for (int q = 0; q < 16; ++q) {
      const int px = ax + (q & 3), py = ay + (q >> 2);
      int v = mode == 0 ? (int)(rnd() % 256) : mode == 1 ? base + (int)(rnd() % 12) : mode == 2 ? ((rnd() % 3) ? base : base + (int)(rnd() % 40)) : (int)(rnd() % 8);
      if (px < 3 || py < 3 || px >= La.w - 3 || py >= La.h - 3) v = 0;  // (k_score_blocks: 0 on the border)
      vals[q] = (uint8_t)brisk_min(v, 255);
    }
    BriskLayerView A = La, B = La;
    brisk_block_from_bytes(&A.blk, vals, 16, ax, ay, 4, 4);
    brisk_block_from_bytes(&B.blk, vals, 16, ax, ay, 4, 4);
    const int thr = mode == 1 || mode == 2 ? base - 8 + (int)(rnd() % 20) : (int)(rnd() % 256);
    BriskTouch ta, tb;
    ta.on = tb.on = true; ta.mask = tb.mask = 0; ta.x0 = tb.x0 = 0; ta.y0 = tb.y0 = 0;
    bool ia = true, ib = true;
    float dxa = 0, dya = 0, dxb = 0, dyb = 0;
    const float ra = brisk_score_max_other<0>(A, true, odd, x, y, thr, ia, dxa, dya, &ta);
    const float rb = brisk_score_max_above_blk(B, odd, x, y, thr, ib, dxb, dyb, &tb);
    const bool same = ia == ib && memcmp(&ra, &rb, 4) == 0 && (!ia || (memcmp(&dxa, &dxb, 4) == 0 && memcmp(&dya, &dyb, 4) == 0)) &&
                      ta.mask == tb.mask && ta.x0 == tb.x0 && ta.y0 == tb.y0 && A.miss == B.miss;
    if (!same) ++bad;
    if (stats) { stats[0] += ia ? 1 : 0; stats[1] += ia ? 0 : 1; stats[2] += A.miss ? 1 : 0; stats[3] += ta.mask ? 1 : 0; }
  }
  return bad;
}

// The same for GetScoreMaxBelow (brisk_score_max_below_blk vs brisk_score_max_other<0>(above = false)): 4 x 4 blocks of
// the layer below with many equal values (the tie rule of :987-1010 moves the maximum between equal inner samples).
int emul_score_max_below_blk_mismatches(unsigned seed, int n, int* stats /* 4: maxima, aborted scans, misses, unused */) {
  auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  int bad = 0;
  for (int it = 0; it < n; ++it) {
    const bool odd = rnd() & 1;
    BriskLayerView Lb;
    memset(&Lb, 0, sizeof(Lb));
    Lb.w = 12 + (int)(rnd() % 80);
    Lb.h = 12 + (int)(rnd() % 80);
    Lb.stride = Lb.w;
    // own layer: 3/4 (even: below is the intra-octave) or 2/3 (odd: below is the octave) of the layer below
    const int ow = odd ? (Lb.w * 2) / 3 : (Lb.w * 3) / 4, oh = odd ? (Lb.h * 2) / 3 : (Lb.h * 3) / 4;
    const int x = 3 + (int)(rnd() % (unsigned)brisk_max(ow - 6, 1)), y = 3 + (int)(rnd() % (unsigned)brisk_max(oh - 6, 1));
    int ax, ay;
    brisk_block_anchor(false, odd, x, y, &ax, &ay);
    uint8_t vals[16];
    const unsigned mode = rnd() % 5;
    const int base = (int)(rnd() % 200);
    for (int q = 0; q < 16; ++q) {
      const int px = ax + (q & 3), py = ay + (q >> 2);
      int v = mode == 0 ? (int)(rnd() % 256) : mode == 1 ? base + (int)(rnd() % 12) : mode == 2 ? ((rnd() % 3) ? base : base + (int)(rnd() % 40))
              : mode == 3 ? base + (int)(rnd() % 3) : (int)(rnd() % 8);
      if (px < 3 || py < 3 || px >= Lb.w - 3 || py >= Lb.h - 3) v = 0;
      vals[q] = (uint8_t)brisk_min(v, 255);
    }
    BriskLayerView A = Lb, B = Lb;
    brisk_block_from_bytes(&A.blk, vals, 16, ax, ay, 4, 4);
    brisk_block_from_bytes(&B.blk, vals, 16, ax, ay, 4, 4);
    const int thr = mode >= 1 && mode <= 3 ? base - 8 + (int)(rnd() % 20) : (int)(rnd() % 256);
    BriskTouch none;
    none.on = false; none.mask = 0; none.x0 = 0; none.y0 = 0;
    bool ia = true, ib = true;
    float dxa = 0, dya = 0, dxb = 0, dyb = 0;
    const float ra = brisk_score_max_other<0>(A, false, odd, x, y, thr, ia, dxa, dya, &none);
    const float rb = brisk_score_max_below_blk(B, odd, x, y, thr, ib, dxb, dyb);
    const bool same = ia == ib && memcmp(&ra, &rb, 4) == 0 && (!ia || (memcmp(&dxa, &dxb, 4) == 0 && memcmp(&dya, &dyb, 4) == 0)) && A.miss == B.miss;
    if (!same) ++bad;
    if (stats) { stats[0] += ia ? 1 : 0; stats[1] += ia ? 0 : 1; stats[2] += A.miss ? 1 : 0; }
  }
  return bad;
}

// The packed form of the OAST 9_16 score (brisk_oast9_16_M_from_pk, k_score_blocks) against brisk_oast9_16_M_from_d on
// random ring differences in [-255, 255] (plateaus and runs included).  Returns mismatches.
int emul_oast_pk_mismatches(unsigned seed, int n) {
  auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  int bad = 0;
  for (int it = 0; it < n; ++it) {
    int d[16];
    const unsigned mode = rnd() % 4;
    const int c = (int)(rnd() % 256);
    for (int j = 0; j < 16; ++j) {
      const int v = mode == 0 ? (int)(rnd() % 256) : mode == 1 ? ((rnd() % 4) ? 200 : (int)(rnd() % 256)) : mode == 2 ? ((rnd() % 4) ? 10 : (int)(rnd() % 256)) : (j < (int)(rnd() % 17) ? 255 : 0);
      d[j] = v - c;
    }
    uint32_t P[8];
    for (int i = 0; i < 8; ++i) P[i] = ((uint32_t)d[i] & 0xFFFFu) | ((uint32_t)d[i + 8] << 16);
    if (brisk_oast9_16_M_from_d(d) != brisk_oast9_16_M_from_pk(P)) ++bad;
  }
  return bad;
}

// brisk_block_anchor uses integer quotients where brisk_score_max_other (and the reference, brisk-scale-space.cc:786-801,
// 946-962) truncates float quotients: identical for every coordinate the engine admits
int emul_block_anchor_mismatches(void) {
  int bad = 0;
  for (int v = 0; v < 8192; ++v) {
    int ax, ay;
    brisk_block_anchor(true, false, v, v, &ax, &ay);
    if (ax != (int)(float)((float)(4 * v - 1 - 2) / 6.0) - 1 || ay != ax) ++bad;
    brisk_block_anchor(true, true, v, v, &ax, &ay);
    if (ax != (int)((float)(6 * v - 1 - 3) / 8.0f) - 1 || ay != ax) ++bad;
    brisk_block_anchor(false, false, v, v, &ax, &ay);
    if (ax != (int)(float)((float)(8 * v + 1 - 4) / 6.0) || ay != ax) ++bad;
    brisk_block_anchor(false, true, v, v, &ax, &ay);
    if (ax != (int)(float)((float)(6 * v + 1 - 3) / 4.0) || ay != ax) ++bad;
  }
  return bad;
}

// k_detect phase A: the packed pre-gate must be a NECESSARY condition of brisk_detect_px.  Walks an image exactly as
// the kernel pairs the pixels (two horizontally adjacent centres per call) and returns the number of detections the
// pre-gate would have dropped (must be 0); *survivors receives the number of pixels that pass it.
int emul_pregate_missed(const uint8_t* img, int w, int h, int stride, int thr, long* survivors) {
  const BriskPregate pg = brisk_pregate_make(thr);
  int missed = 0;
  long pass = 0;
  auto px = [&](int x, int y) -> uint32_t { return img[(long)y * stride + x]; };
  for (int y = 3; y <= h - 4; ++y)
    for (int x = 3; x <= w - 4; x += 2) {
      const int x1 = (x + 1 <= w - 4) ? x + 1 : x;  // second lane of the pair (duplicate of the first at the edge)
      auto pair = [&](int dx, int dy) { return px(x + dx, y + dy) | (px(x1 + dx, y + dy) << 16); };
      const uint32_t g = brisk_pregate_pair(pair(0, 0), pair(0, -3), pair(0, 3), pair(-3, 0), pair(3, 0), pg);
      const int lanes[2] = {x, x1};
      for (int k = 0; k < 2; ++k) {
        if (k == 1 && x1 == x) break;
        const bool gate = ((g >> (16 * k)) & 0xFFFFu) != 0;
        pass += gate;
        if (!gate && brisk_detect_px(img + (long)y * stride + lanes[k], stride, thr)) ++missed;
      }
    }
  if (survivors) *survivors = pass;
  return missed;
}
// the pre-gate's lower bound of the adaptive threshold: (tc * K) >> s <= (tc * thr) / 100 and no 16-bit overflow
int emul_pregate_bound_violations(void) {
  int bad = 0;
  for (int thr = 1; thr <= 255; ++thr) {
    const BriskPregate pg = brisk_pregate_make(thr);
    const unsigned K = pg.K & 0xFFFFu, s = pg.shift & 0xFFFFu;
    if ((pg.K >> 16) != K || (pg.shift >> 16) != s) ++bad;
    for (unsigned tc = 10; tc <= 230; ++tc) {
      if (tc * K > 65535u) ++bad;
      if ((int)((tc * K) >> s) > (int)(tc * thr) / 100) ++bad;
    }
  }
  return bad;
}

// closed-form scores vs the oracle's bisection (per pixel)
int emul_oast_Kp(const uint8_t* p, int stride) { return brisk_Kp_from_M(brisk_oast9_16_M(p, stride)); }
int emul_agast58_Kp(const uint8_t* p, int stride) { return brisk_Kp_from_M(brisk_agast5_8_M(p, stride)); }
int emul_detect_px(const uint8_t* p, int stride, int thr) { return brisk_detect_px(p, stride, thr); }

void* emul_pattern_create(int version, float pattern_scale, const char* text) {
  EmulPattern* e = new EmulPattern();
  std::string err;
  const bool ok = text ? brisk_pattern_build_from_text(text, pattern_scale, &e->H, &err)
                       : brisk_pattern_build_default(version, pattern_scale, &e->H, &err);
  if (!ok) { delete e; return nullptr; }
  BriskPatternDev& d = e->P;
  d.npoints = e->H.npoints; d.nshort = e->H.nshort; d.nlong = e->H.nlong; d.strings = e->H.strings;
  d.rotation_invariant = 1; d.scale_invariant = 1; d.basicscale = e->H.basicscale;
  d.mult = e->H.mult.data(); d.sigma = e->H.sigma.data(); d.uv = e->H.uv.data(); d.scaling = e->H.scaling.data();
  d.size_thresh = e->H.size_thresh.data(); d.size_list = e->H.size_list.data();
  d.short_pairs = e->H.short_pairs.data(); d.long_pairs = e->H.long_pairs.data();
  return e;
}
void emul_pattern_destroy(void* p) { delete (EmulPattern*)p; }
int emul_pattern_strings(void* p) { return ((EmulPattern*)p)->H.strings; }
int emul_pattern_points(void* p) { return ((EmulPattern*)p)->H.npoints; }
void emul_pattern_tables(void* p, float* scale_list, int* size_list, float* size_thresh) {
  EmulPattern* e = (EmulPattern*)p;
  memcpy(scale_list, e->H.scale_list.data(), 256);
  memcpy(size_list, e->H.size_list.data(), 256);
  memcpy(size_thresh, e->H.size_thresh.data(), 256);
}
// full pattern LUT entry as the reference tabulates it
void emul_pattern_point(void* p, int scale, int rot, int i, float* xyz) {
  const BriskSamplePoint sp = brisk_pattern_point(((EmulPattern*)p)->P, scale, rot, i);
  xyz[0] = sp.x; xyz[1] = sp.y; xyz[2] = sp.sigma;
}
int emul_scale_index(void* p, float size, int scale_invariant) {
  BriskPatternDev P = ((EmulPattern*)p)->P;
  P.scale_invariant = scale_invariant;
  return brisk_scale_index(P, size);
}
int emul_scale_index_host(float size) { return brisk_pattern_scale_index_host(size); }

// brisk_div_by_magic (k_describe's division by scaling2) against C division: every scaling2 of the pattern's table, the
// given numerators plus edge values; returns the number of mismatches
long emul_div_magic_mismatches(void* p, const int* numerators, int n) {
  const BriskPatternHost& H = ((EmulPattern*)p)->H;
  long bad = 0;
  const int edge[] = {0, 1, -1, 2147483647, -2147483647 - 1, -2147483647, 1073741824, 4095, 4096, 4097, -4096, 65535};
  for (size_t i = 0; i < H.scaling.size() / 2; ++i) {
    const int d = H.scaling[2 * i + 1];
    if (d < 2) continue;
    int M, sh;
    brisk_div_magic(d, &M, &sh);
    for (int k = 0; k < n; ++k) bad += (numerators[k] / d != brisk_div_by_magic(numerators[k], M, sh));
    for (int e : edge) {
      bad += (e / d != brisk_div_by_magic(e, M, sh));
      bad += ((e - d) / d != brisk_div_by_magic(e - d, M, sh)) + ((d * 3 + e % 7) / d != brisk_div_by_magic(d * 3 + e % 7, M, sh));
    }
  }
  return bad;
}
// the same for an arbitrary divisor
long emul_div_magic_mismatches_d(int d, const int* numerators, int n) {
  int M, sh;
  brisk_div_magic(d, &M, &sh);
  long bad = 0;
  for (int k = 0; k < n; ++k) bad += (numerators[k] / d != brisk_div_by_magic(numerators[k], M, sh));
  return bad;
}

// one sample the way k_describe computes it: parameters from the packed table record (brisk_pack_tab) and the unit
// offset, box corners / weights (brisk_box_prep), the 4 x 4 integral samples + the two displaced corner pixels, the
// weighted sum (brisk_box_acc), the division by multiplication (brisk_box_divide); points on the bilinear branch through
// the generic function (the kernel's GENERIC variant)
// every sample of emul_describe also goes through the two-sided form of the box sum (brisk_box_side / brisk_box_acc_pair:
// what k_describe's lane pairs compute since round 5), as the even and as the odd lane, on 32-bit and on 24-bit samples
static long g_pair_checked = 0, g_pair_mismatch = 0;
extern "C" void emul_box_pair_stats(long* checked, long* mismatches) { *checked = g_pair_checked; *mismatches = g_pair_mismatch; }
static int emul_sample(const BriskPatternDev& P, const uint8_t* img, int stride, int cols, const uint32_t* integ, int istride, float kx,
                       float ky, int scale, int theta, int i) {
  const int ti = scale * P.npoints + i;
  const float sigma = P.sigma[ti];
  if (sigma < 0.5f) return brisk_smoothed_intensity(img, stride, cols, integ, istride, kx, ky, brisk_pattern_point(P, scale, theta, i));
  int tz, tw;
  brisk_pack_tab(P.scaling[2 * ti], P.scaling[2 * ti + 1], &tz, &tw);
  const double mm = (double)P.mult[ti];
  const double* uv = P.uv + ((long)theta * P.npoints + i) * 2;
  const float xf = (float)(mm * uv[0]) + kx, yf = (float)(mm * uv[1]) + ky;
  const BriskBoxPrep p = brisk_box_prep(xf, yf, sigma, tz, tw);
  const uint32_t* r0 = integ + (long)p.y_top * istride;
  const uint32_t* r1 = r0 + istride;
  const uint32_t* r2 = integ + (long)p.y_bottom * istride;
  const uint32_t* r3 = r2 + istride;
  const int qy = p.y_bottom - 1 > 0 ? p.y_bottom - 1 : 0;
  const unsigned qbr = brisk_linear_px(img, stride, cols, p.x_right + 1, qy), qbl = brisk_linear_px(img, stride, cols, p.x_left + 1, qy);
  const uint32_t acc = brisk_box_acc(p, r0[p.x_left], r0[p.x_left + 1], r0[p.x_right], r0[p.x_right + 1], r1[p.x_left], r1[p.x_left + 1],
                                     r1[p.x_right], r1[p.x_right + 1], r2[p.x_left], r2[p.x_left + 1], r2[p.x_right], r2[p.x_right + 1],
                                     r3[p.x_left], r3[p.x_left + 1], r3[p.x_right], r3[p.x_right + 1], qbr, qbl);
  {
    const bool wrap = p.quirk && (p.x_right + 1 >= cols || p.x_left + 1 >= cols);  // (there the kernel reads the frame)
    const uint32_t* rq = integ + (long)qy * istride;
    for (int m24 = 0; m24 < 2; ++m24) {
      const uint32_t mask = m24 ? 0xFFFFFFu : 0xFFFFFFFFu;
      auto V = [&](const uint32_t* r, int x) { return r[x] & mask; };
      BriskBoxSide Ls = brisk_box_side(V(r0, p.x_left), V(r0, p.x_left + 1), V(r1, p.x_left), V(r1, p.x_left + 1), V(rq, p.x_left + 1),
                                       V(rq, p.x_left + 2), V(r2, p.x_left), V(r2, p.x_left + 1), V(r2, p.x_left + 2), V(r3, p.x_left),
                                       V(r3, p.x_left + 1), p.quirk, false, mask);
      BriskBoxSide Rs = brisk_box_side(V(r0, p.x_right), V(r0, p.x_right + 1), V(r1, p.x_right), V(r1, p.x_right + 1), V(rq, p.x_right + 1),
                                       V(rq, p.x_right + 2), V(r2, p.x_right), V(r2, p.x_right + 1), V(r2, p.x_right + 2), V(r3, p.x_right),
                                       V(r3, p.x_right + 1), p.quirk, true, mask);
      if (wrap) { Ls.cb = qbl; Rs.cb = qbr; }
      const uint32_t want = brisk_box_acc(p, V(r0, p.x_left), V(r0, p.x_left + 1), V(r0, p.x_right), V(r0, p.x_right + 1), V(r1, p.x_left),
                                          V(r1, p.x_left + 1), V(r1, p.x_right), V(r1, p.x_right + 1), V(r2, p.x_left), V(r2, p.x_left + 1),
                                          V(r2, p.x_right), V(r2, p.x_right + 1), V(r3, p.x_left), V(r3, p.x_left + 1), V(r3, p.x_right),
                                          V(r3, p.x_right + 1), qbr, qbl, mask);
      g_pair_checked += 2;
      if (brisk_box_acc_pair(p, Ls, Rs, false, mask) != want) ++g_pair_mismatch;
      if (brisk_box_acc_pair(p, Rs, Ls, true, mask) != want) ++g_pair_mismatch;
    }
  }
  return brisk_box_divide(p, acc);
}

// mirrors k_integral_* + k_desc_prepare + k_describe; returns surviving count
int emul_describe(void* pat, const uint8_t* img, int w, int h, BriskKeyPoint* kps, int n, uint8_t* desc, int desc_pitch,
                  int rotation_invariant, int scale_invariant) {
  BriskPatternDev P = ((EmulPattern*)pat)->P;
  P.rotation_invariant = rotation_invariant;
  P.scale_invariant = scale_invariant;
  const int istride = brisk_align_up(w + 1, 16);
  // the image in the engine's layout: rows padded to a multiple of 64 bytes (a sampling function that assumed
  // stride == cols would read the padding instead of the next row)
  const int stride = brisk_align_up(w, BRISK_STRIDE_ALIGN);
  std::vector<uint8_t> padded((size_t)stride * h + 256, 0xEE);
  for (int y = 0; y < h; ++y) memcpy(padded.data() + (size_t)y * stride, img + (size_t)y * w, w);
  std::vector<uint32_t> integ((size_t)istride * (h + 1), 0);
  for (int y = 0; y < h; ++y) {
    uint32_t s = 0;
    for (int x = 0; x < w; ++x) {
      s += img[(size_t)y * w + x];
      integ[(size_t)(y + 1) * istride + x + 1] = integ[(size_t)y * istride + x + 1] + s;
    }
  }
  std::vector<int> dscale;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const int sc = brisk_scale_index(P, kps[i].size);
    if (brisk_inside_border(P, sc, kps[i].x, kps[i].y, w, h)) { kps[m++] = kps[i]; dscale.push_back(sc); }
  }
  std::vector<int> values(BRISK_MAX_POINTS);
  for (int k = 0; k < m; ++k) {
    BriskKeyPoint* kp = &kps[k];
    const int scale = dscale[k];
    int theta = 0;
    if (P.rotation_invariant) {
      if (kp->angle == -1.0f) {
        for (int i = 0; i < P.npoints; ++i) {
          values[i] = emul_sample(P, padded.data(), stride, w, integ.data(), istride, kp->x, kp->y, scale, 0, i);
        }
        int d0 = 0, d1 = 0;
        for (int p = 0; p < P.nlong; ++p) {
          int a, b;
          brisk_long_pair(values.data(), P.long_pairs + 4 * p, &a, &b);
          d0 += a; d1 += b;
        }
        kp->angle = brisk_angle_from_direction(d0, d1);
        theta = brisk_theta_from_angle(kp->angle, true);
      } else {
        theta = brisk_theta_from_angle(kp->angle, false);
      }
    }
    for (int i = 0; i < P.npoints; ++i) {
      values[i] = emul_sample(P, padded.data(), stride, w, integ.data(), istride, kp->x, kp->y, scale, theta, i);
    }
    uint8_t* drow = desc + (size_t)k * desc_pitch;
    memset(drow, 0, P.strings);
    for (int p = 0; p < P.nshort; ++p)
      if (values[P.short_pairs[2 * p]] > values[P.short_pairs[2 * p + 1]]) drow[p >> 3] |= (uint8_t)(1u << (p & 7));
  }
  return m;
}

}  // extern "C"
