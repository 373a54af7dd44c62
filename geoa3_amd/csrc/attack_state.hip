// Device-resident state machine of attack() (Attacker/geoA3_attack.py:182-386): classification loss,
// success bookkeeping, optimiser step and binary search, with no host synchronisation.
#include "common.h"

namespace {

constexpr int HEAD_BLOCK = 256;

// One workgroup per instance.  Wave 0 evaluates the classification loss and its gradient w.r.t. the
// logits (geoA3_attack.py:105-127), lane 0 then runs the bookkeeping of geoA3_attack.py:297-310 and the
// whole workgroup copies the iterate into best_attack when it improved.
// vote_logits (optional, [B, eval_num, classes]): the dense-cloud success check of geoA3_attack.py:289-295 -- the
// instance succeeds when MORE than half of its eval_num resampled clouds satisfy _compare, and its output_label is
// the mode of their arg-max labels (torch.mode: the smallest of equally frequent values).
__global__ __launch_bounds__(HEAD_BLOCK) void attack_head_kernel(geoa3_attack_state st, const float* __restrict__ logits,
                                                                 const float* __restrict__ vote_logits, int eval_num,
                                                                 const float* __restrict__ constrain,
                                                                 const float* __restrict__ x, int step,
                                                                 int search_step, float* __restrict__ dlogits,
                                                                 int phase, int32_t* __restrict__ okbuf) {
  // phase 0: everything.  phase 1: the classification part only (loss, d loss / d logits, label, success flag ->
  // okbuf) -- it does not need the constrain loss, so the victim's backward can start while the geometry kernels still
  // run; phase 2: the bookkeeping of geoA3_attack.py:297-310 from what phase 1 stored, once the constrain loss is there.
  __shared__ int s_copy;
  __shared__ int s_vote[GEOA3_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x, C = st.classes;
  const float* lg = logits + (size_t)b * C;
  if (phase == 2) {
    if (tid == 0) {
      const float cls = st.cls_loss[b];
      const int am = st.label[b];
      const bool ok = okbuf[b] != 0;
      const float con = constrain ? constrain[b] : 0.f;
      const float ln = cls + st.scale_const[b] * con;
      st.loss_n[b] = ln;
      if (st.loss_hist) st.loss_hist[(size_t)step * st.B + b] = ln;
      const float metric = st.prev_constrain[b];
      int copy = 0;
      if (ok && metric < st.best_loss[b]) {
        st.best_loss[b] = metric;
        st.best_step[b] = step;
        st.best_bs[b] = search_step;
        copy = 1;
      }
      if (ok && metric < st.iter_best_loss[b]) {
        st.iter_best_loss[b] = metric;
        st.iter_best_score[b] = am;
      }
      st.prev_constrain[b] = con;
      s_copy = copy;
    }
  } else if (tid < GEOA3_WAVE) {
    const int lane = tid;
    const int tgt = st.target[b];
    // arg-max (first maximal index, as torch.argmax) and max over c != target
    float mx = -__builtin_inff();
    int am = 0x7fffffff;
    float other = -__builtin_inff();
    int oi = 0x7fffffff;
    for (int c = lane; c < C; c += GEOA3_WAVE) {
      const float v = lg[c];
      if (v > mx) { mx = v; am = c; }
      const float w = (c == tgt) ? -10000.0f : v;  // (1-onehot)*logits - onehot*1e4, geoA3_attack.py:110
      if (w > other) { other = w; oi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(mx, o, 64);
      const int i2 = __shfl_xor(am, o, 64);
      if (v2 > mx || (v2 == mx && i2 < am)) { mx = v2; am = i2; }
      const float w2 = __shfl_xor(other, o, 64);
      const int j2 = __shfl_xor(oi, o, 64);
      if (w2 > other || (w2 == other && j2 < oi)) { other = w2; oi = j2; }
    }
    float cls = 0.f;
    const float invB = st.inv_global_batch;
    if (st.cls_loss_type == 1) {  // CE (reduction none); untargeted: -CE
      float se = 0.f;
      for (int c = lane; c < C; c += GEOA3_WAVE) se += expf(lg[c] - mx);
      se = wave_sum(se);
      const float lse = logf(se) + mx;
      const float ce = lse - lg[tgt];
      const float sgn = st.targeted ? 1.f : -1.f;
      cls = sgn * ce;
      for (int c = lane; c < C; c += GEOA3_WAVE) {
        const float p = expf(lg[c] - lse);
        dlogits[(size_t)b * C + c] = sgn * invB * (p - (c == tgt ? 1.f : 0.f));
      }
    } else if (st.cls_loss_type == 2) {  // Margin
      const float fake = lg[tgt];
      const float raw = st.targeted ? (other - fake + st.confidence) : (fake - other + st.confidence);
      cls = fmaxf(raw, 0.f);
      const float gact = raw >= 0.f ? invB : 0.f;  // clamp(min=0) passes the gradient at equality
      const float g_other = st.targeted ? gact : -gact;
      for (int c = lane; c < C; c += GEOA3_WAVE) {
        float g = 0.f;
        if (c == oi) g += g_other;
        if (c == tgt) g -= g_other;
        dlogits[(size_t)b * C + c] = g;
      }
    } else {
      for (int c = lane; c < C; c += GEOA3_WAVE) dlogits[(size_t)b * C + c] = 0.f;
    }
    bool ok = st.targeted ? (am == tgt) : (am != st.gt[b]);
    if (vote_logits) {
      int n_ok = 0;
      for (int e = 0; e < eval_num; ++e) {
        const float* vl = vote_logits + ((size_t)b * eval_num + e) * C;
        float vm = -__builtin_inff();
        int va = 0x7fffffff;
        for (int c = lane; c < C; c += GEOA3_WAVE) {
          const float v = vl[c];
          if (v > vm) { vm = v; va = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float v2 = __shfl_xor(vm, o, 64);
          const int i2 = __shfl_xor(va, o, 64);
          if (v2 > vm || (v2 == vm && i2 < va)) { vm = v2; va = i2; }
        }
        n_ok += (st.targeted ? (va == tgt) : (va != st.gt[b])) ? 1 : 0;
        if (lane == 0) s_vote[e] = va;
      }
      // mode of the eval_num labels: lane e counts the occurrences of its own label
      int cnt = 0, lab = 0x7fffffff;
      if (lane < eval_num) {
        lab = s_vote[lane];
        for (int e = 0; e < eval_num; ++e) cnt += (s_vote[e] == lab) ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int c2 = __shfl_xor(cnt, o, 64);
        const int l2 = __shfl_xor(lab, o, 64);
        if (c2 > cnt || (c2 == cnt && l2 < lab)) { cnt = c2; lab = l2; }
      }
      am = lab;
      ok = (float)n_ok > 0.5f * (float)eval_num;
    }
    if (lane == 0 && phase == 1) {
      st.cls_loss[b] = cls;
      st.label[b] = am;
      if (b == st.B - 1) *st.last_label = am;
      okbuf[b] = ok ? 1 : 0;
      s_copy = 0;
    } else if (lane == 0) {
      const float con = constrain ? constrain[b] : 0.f;
      const float ln = cls + st.scale_const[b] * con;
      st.cls_loss[b] = cls;
      st.loss_n[b] = ln;
      if (st.loss_hist) st.loss_hist[(size_t)step * st.B + b] = ln;
      st.label[b] = am;
      if (b == st.B - 1) *st.last_label = am;
      const float metric = st.prev_constrain[b];  // constrain of the PREVIOUS step (1e10 at step 0)
      int copy = 0;
      if (ok && metric < st.best_loss[b]) {
        st.best_loss[b] = metric;
        st.best_step[b] = step;
        st.best_bs[b] = search_step;
        copy = 1;
      }
      if (ok && metric < st.iter_best_loss[b]) {
        st.iter_best_loss[b] = metric;
        st.iter_best_score[b] = am;
      }
      st.prev_constrain[b] = con;
      s_copy = copy;
    }
  }
  __syncthreads();
  if (s_copy) {
    const size_t n3 = (size_t)3 * st.N;
    const float* src = x + (size_t)b * n3;
    float* dst = st.best_attack + (size_t)b * n3;
    for (size_t i = tid; i < n3; i += HEAD_BLOCK) dst[i] = src[i];
  }
}

// One thread per point (all three coordinates, so lp_clip sees the whole offset vector).
__global__ __launch_bounds__(256) void attack_update_kernel(int B, int N, const float* __restrict__ scale_const,
                                                            float inv_global_batch, const float* __restrict__ g_cls,
                                                            const float* __restrict__ g_geo,
                                                            const float* __restrict__ ori, float* __restrict__ offset,
                                                            float* __restrict__ am, float* __restrict__ av,
                                                            float* __restrict__ x, int optim, float step_size,
                                                            float sqrt_bc2, float cc_linf) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)B * N) return;
  const int b = (int)(p / N);
  const int n = (int)(p - (size_t)b * N);
  const float cg = g_geo ? scale_const[b] * inv_global_batch : 0.f;
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t e = ((size_t)b * 3 + c) * N + n;
    float g = g_cls ? g_cls[e] : 0.f;
    if (g_geo) g += cg * g_geo[e];
    float w = offset[e];
    if (optim == 0) {  // torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8
      // torch forms 1 - beta in DOUBLE and rounds once: 0.1f and 0.001f (1.0f - 0.999f in float is 0.00099998713:
      // 1.3e-5 off in v, 6.5e-6 in the step -- found by the adam/* reference trace, tests/test_gpu_forward_step.py).
      // The trace was recorded with torch 2.10.0 (CPU, the single-tensor path: exp_avg.lerp_(grad, 1 - beta1), i.e.
      // m + 0.1 (g - m); exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)); the form below agrees with it to the
      // bar the test states (2 ulp of the iterate + 3e-7 of the step), not bit for bit -- older torch (mul_/add_) rounds
      // m differently in the last place, so no bit claim is made for any version.
      const float m = am[e] * 0.9f + g * 0.1f;
      const float v = av[e] * 0.999f + (g * g) * 0.001f;
      am[e] = m;
      av[e] = v;
      const float denom = sqrtf(v) / sqrt_bc2 + 1e-8f;
      w = w - step_size * (m / denom);
    } else {  // plain SGD (geoA3_attack.py:271-272)
      w = w - step_size * g;
    }
    o[c] = w;
  }
  if (cc_linf != 0.f) {  // lp_clip, geoA3_attack.py:88-98
    const float len = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    if (!(len < cc_linf)) {
      const bool big = len > 1e-6f;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = big ? o[c] / len * cc_linf : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t e = ((size_t)b * 3 + c) * N + n;
    offset[e] = o[c];
    x[e] = ori[e] + o[c];
  }
}

__global__ __launch_bounds__(256) void attack_project_kernel(int mode, int B, int N, const float* __restrict__ ori,
                                                             const float* __restrict__ normal,
                                                             const int32_t* __restrict__ nn, float* __restrict__ offset,
                                                             float* __restrict__ x, float cc_linf) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)B * N) return;
  const int b = (int)(p / N);
  const int n = (int)(p - (size_t)b * N);
  const size_t base = (size_t)b * 3 * N;
  const int j = nn[p];
  if (mode == 0) {  // find_offset: measured from the nearest original point
#pragma unroll
    for (int c = 0; c < 3; ++c) offset[base + (size_t)c * N + n] = x[base + (size_t)c * N + n] - ori[base + (size_t)c * N + j];
    return;
  }
  float o[3], u[3];
  float n2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[c] = offset[base + (size_t)c * N + n];
    u[c] = normal[base + (size_t)c * N + j];
    n2 += u[c] * u[c];
  }
  const float den = sqrtf(n2) + 1e-6f;
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    u[c] = u[c] / den;
    dot += o[c] * u[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = dot * u[c];
  if (cc_linf != 0.f) {
    const float len = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    if (!(len < cc_linf)) {
      const bool big = len > 1e-6f;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = big ? o[c] / len * cc_linf : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t e = base + (size_t)c * N + n;
    offset[e] = o[c];
    x[e] = ori[e] + o[c];
  }
}

__global__ void binary_update_kernel(geoa3_attack_state st) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= st.B) return;
  const int lab = *st.last_label;  // geoA3_attack.py:375 uses the last instance's label for every k
  const bool cmp = st.targeted ? (lab == st.target[k]) : (lab != st.gt[k]);
  float c = st.scale_const[k], lo = st.lower_bound[k], up = st.upper_bound[k];
  if (cmp && st.iter_best_score[k] != -1) {
    lo = fmaxf(lo, c);
    if (up < 1e9f) c = (lo + up) * 0.5f;
    else c = c * 2.0f;
  } else {
    up = fminf(up, c);
    if (up < 1e9f) c = (lo + up) * 0.5f;
  }
  st.scale_const[k] = c;
  st.lower_bound[k] = lo;
  st.upper_bound[k] = up;
}

__global__ __launch_bounds__(256) void begin_search_step_kernel(geoa3_attack_state st, const float* __restrict__ ori,
                                                                const float* __restrict__ init_offset,
                                                                float* __restrict__ offset, float* __restrict__ am,
                                                                float* __restrict__ av, float* __restrict__ x) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)st.B * 3 * st.N;
  if (e < total) {
    const float o = init_offset[e];
    offset[e] = o;
    if (am) am[e] = 0.f;
    if (av) av[e] = 0.f;
    x[e] = ori[e] + o;
  }
  if (e < (size_t)st.B) {
    st.iter_best_loss[e] = 1e10f;
    st.iter_best_score[e] = -1;
    st.prev_constrain[e] = 1e10f;
  }
}

// --is_partial_var (geoA3_attack.py:239-262,278-281): the optimised variable is a [B,3,kr] offset on kr points of every
// instance.  One thread per (instance, point): gather the full gradient, optimiser step, x = periodical + offset there.
__global__ __launch_bounds__(256) void attack_partial_kernel(int B, int N, int kr, const float* __restrict__ scale_const,
                                                             float inv_global_batch, const float* __restrict__ g_cls,
                                                             const float* __restrict__ g_geo,
                                                             const int32_t* __restrict__ pidx,
                                                             const float* __restrict__ periodical,
                                                             float* __restrict__ part, float* __restrict__ pm,
                                                             float* __restrict__ pv, float* __restrict__ x, int optim,
                                                             float step_size, float sqrt_bc2, float momentum, int first) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= B * kr) return;
  const int b = p / kr, j = p - b * kr;
  const int i = pidx[p];
  const float cg = g_geo ? scale_const[b] * inv_global_batch : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t e = ((size_t)b * 3 + c) * N + i, pe = ((size_t)b * 3 + c) * kr + j;
    float w = part[pe];
    if (optim >= 0) {
      float g = g_cls ? g_cls[e] : 0.f;
      if (g_geo) g += cg * g_geo[e];
      if (optim == 0) {   // torch.optim.Adam defaults
        const float m = pm[pe] * 0.9f + g * 0.1f;            // 1 - beta rounded from double, as torch (see above)
        const float v = pv[pe] * 0.999f + (g * g) * 0.001f;
        pm[pe] = m;
        pv[pe] = v;
        w = w - step_size * (m / (sqrtf(v) / sqrt_bc2 + 1e-8f));
      } else {            // torch.optim.SGD(momentum), geoA3_attack.py:252: buf = g at the first step
        const float buf = first ? g : pm[pe] * momentum + g;
        pm[pe] = buf;
        w = w - step_size * buf;
      }
      part[pe] = w;
    }
    x[e] = periodical[e] + w;
  }
}

}  // namespace

extern "C" int geoa3_attack_partial_step(const geoa3_attack_state* st, const float* g_cls, const float* g_geo,
                                         const int32_t* pidx, int kr, const float* periodical, float* part, float* pm,
                                         float* pv, float* x, int optim, float step_size, float sqrt_bc2, float momentum,
                                         int first, void* stream) {
  if (!st || !pidx || !periodical || !part || !x || kr <= 0 || optim < -1 || optim > 1) return GEOA3_EINVAL;
  if (optim >= 0 && (!pm || (optim == 0 && !pv))) return GEOA3_EINVAL;
  const int total = st->B * kr;
  hipLaunchKernelGGL(attack_partial_kernel, dim3((total + 255) / 256), dim3(256), 0, geoa3_stream(stream), st->B, st->N,
                     kr, st->scale_const, st->inv_global_batch, g_cls, g_geo, pidx, periodical, part, pm, pv, x, optim,
                     step_size, sqrt_bc2, momentum, first);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_head(const geoa3_attack_state* st, const float* logits, const float* constrain,
                                 const float* x, int step, int search_step, float* dlogits, void* stream) {
  return geoa3_attack_head_vote(st, logits, nullptr, 0, constrain, x, step, search_step, dlogits, stream);
}

extern "C" int geoa3_attack_head_vote(const geoa3_attack_state* st, const float* logits, const float* vote_logits,
                                      int eval_num, const float* constrain, const float* x, int step, int search_step,
                                      float* dlogits, void* stream) {
  if (!st || !logits || !x || !dlogits || st->B <= 0 || st->classes <= 0) return GEOA3_EINVAL;
  if (vote_logits && (eval_num <= 0 || eval_num > GEOA3_WAVE)) return GEOA3_EINVAL;
  hipLaunchKernelGGL(attack_head_kernel, dim3(st->B), dim3(HEAD_BLOCK), 0, geoa3_stream(stream), *st, logits,
                     vote_logits, eval_num, constrain, x, step, search_step, dlogits, 0, (int32_t*)nullptr);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_head_classify(const geoa3_attack_state* st, const float* logits, const float* vote_logits,
                                          int eval_num, float* dlogits, int32_t* ok, void* stream) {
  if (!st || !logits || !dlogits || !ok || st->B <= 0 || st->classes <= 0) return GEOA3_EINVAL;
  if (vote_logits && (eval_num <= 0 || eval_num > GEOA3_WAVE)) return GEOA3_EINVAL;
  hipLaunchKernelGGL(attack_head_kernel, dim3(st->B), dim3(HEAD_BLOCK), 0, geoa3_stream(stream), *st, logits,
                     vote_logits, eval_num, (const float*)nullptr, (const float*)nullptr, 0, 0, dlogits, 1, ok);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_head_finish(const geoa3_attack_state* st, const int32_t* ok, const float* constrain,
                                        const float* x, int step, int search_step, void* stream) {
  if (!st || !ok || !x || st->B <= 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(attack_head_kernel, dim3(st->B), dim3(HEAD_BLOCK), 0, geoa3_stream(stream), *st,
                     (const float*)nullptr, (const float*)nullptr, 0, constrain, x, step, search_step, (float*)nullptr, 2,
                     const_cast<int32_t*>(ok));
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_update(const geoa3_attack_state* st, const float* g_cls, const float* g_geo,
                                   const float* ori, float* offset, float* adam_m, float* adam_v, float* x,
                                   int optim, float step_size, float sqrt_bc2, float cc_linf, void* stream) {
  if (!st || !ori || !offset || !x) return GEOA3_EINVAL;
  if (optim == 0 && (!adam_m || !adam_v)) return GEOA3_EINVAL;
  const size_t pts = (size_t)st->B * st->N;
  hipLaunchKernelGGL(attack_update_kernel, dim3((unsigned)((pts + 255) / 256)), dim3(256), 0, geoa3_stream(stream),
                     st->B, st->N, st->scale_const, st->inv_global_batch, g_cls, g_geo, ori, offset, adam_m, adam_v,
                     x, optim, step_size, sqrt_bc2, cc_linf);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_project(int mode, const float* ori, const float* normal_ori, const int32_t* nn,
                                    float* offset, float* x, int B, int N, float cc_linf, void* stream) {
  if (!ori || !nn || !offset || !x || B <= 0 || N <= 0 || (mode != 0 && mode != 1)) return GEOA3_EINVAL;
  if (mode == 1 && !normal_ori) return GEOA3_EINVAL;
  const size_t pts = (size_t)B * N;
  hipLaunchKernelGGL(attack_project_kernel, dim3((unsigned)((pts + 255) / 256)), dim3(256), 0, geoa3_stream(stream),
                     mode, B, N, ori, normal_ori, nn, offset, x, cc_linf);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_binary_update(const geoa3_attack_state* st, void* stream) {
  if (!st || st->B <= 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(binary_update_kernel, dim3((st->B + 255) / 256), dim3(256), 0, geoa3_stream(stream), *st);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_attack_begin_search_step(const geoa3_attack_state* st, const float* ori,
                                              const float* init_offset, float* offset, float* adam_m, float* adam_v,
                                              float* x, void* stream) {
  if (!st || !ori || !init_offset || !offset || !x) return GEOA3_EINVAL;
  const size_t total = (size_t)st->B * 3 * st->N;
  hipLaunchKernelGGL(begin_search_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     geoa3_stream(stream), *st, ori, init_offset, offset, adam_m, adam_v, x);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_version(void) { return GEOA3_ABI_VERSION; }

extern "C" const char* geoa3_strerror(int code) {
  switch (code) {
    case GEOA3_OK: return "ok";
    case GEOA3_EINVAL: return "invalid argument";
    case GEOA3_ELAUNCH: return "HIP kernel launch failed";
    case GEOA3_ENOSUPPORT: return "size outside the supported range";
    default: return "unknown error";
  }
}
