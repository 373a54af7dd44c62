// Minimal, self-contained reproducer of NOTEBOOK 5a (no library, nothing but hipcc; gfx950 / MI355X, ROCm 7.2):
//     v_pk_add_f32 d, p, A op_sel:[0,1]          (d.lo = p.lo + A.HI, d.hi = p.hi + A.hi)
// -- packed FP32 arithmetic whose src1 carries an op_sel bit, i.e. a LOW result that reads the HIGH half of a register pair --
// is exact alone and returns, in lanes 48-63 of the wavefront, the result of an operand that reads as ZERO (add: p.lo
// unchanged; mul: -0; fma: the addend) when the wavefronts that share its SIMDs interleave independent vector fma with
// v_mfma_f32_32x32x16_f16.  The same with v_pk_mul_f32 / v_pk_fma_f32; neg modifiers, an LDS read in flight, op_sel on src0,
// op_sel_hi and v_pk_mov_b32 do not matter / are clean.  Every packed result is checked against scalar instructions in the
// same lane; mismatches are counted and the first ones printed with their operands.
//   hipcc --offload-arch=gfx950 -O3 -o pk_neg_mfma_min pk_neg_mfma_min.hip && ./pk_neg_mfma_min      (PK_ALL=1: + the
//   single-instruction victims 0-12 that never failed)          output: profiles/round5_pk_opsel_min_reproducer.txt
// (The file's name records the first hypothesis -- neg modifiers beside matrix instructions -- which this program refuted.)
// VICTIM 13-22 run a three-instruction sequence as the farthest-point sampler's round has it: ds_read2_b32 v[100:101] (A),
// ds_read_b32 v102 (B), then  d0 = p - A.lo (op_sel_hi:[1,0]),  d1 = <the instruction under test> on A,  d2 = p3 - B.lo:
//   13: d1 = v_pk_add_f32 p2, A op_sel:[0,1] neg (the sampler's, the second LDS read still in flight under d0 / d1)
//   14: as 13, all LDS data landed first     15: as 13 without neg     16: op_sel_hi:[1,0] on all three (no op_sel)
//   17: v_pk_mov_b32 op_sel:[1,0] (swap)     18: v_pk_mul_f32 op_sel:[0,1]     19: v_pk_fma_f32 op_sel:[0,1,0]
//   20: v_pk_add_f32 A, p2 op_sel:[1,0] (the pair as src0)     21 / 22: v_pk_mul_f32 A, p2 op_sel:[1,0] (/ + op_sel_hi:[0,1])
//   23: v_fma_mixhi_f16 ... op_sel:[0,0,1] op_sel_hi:[0,0,1] (an f16 operand from the HIGH 16 bits of one register)
// VICTIM 0-12: one packed instruction per iteration, earlier hypotheses (none ever failed):
// VICTIM 0: the packed subtraction with neg modifiers; 1: the packed ADD of a pre-negated q (no modifier); 2: v_pk_mul_f32
// with neg; 3: v_pk_fma_f32 with neg on the addend; 4: as 0 with q read from LDS (ds_read2_b32, every lane the same address)
// right in front of the packed instruction, as the sampler does; 5: as 4 WITHOUT a modifier (d = p + q); 6: as 4 with eight
// idle cycles between the LDS wait and the packed instruction; 8: as 4 behind a workgroup barrier (one per iteration, as the
// sampler's rounds) and with the sampler's op_sel (q's low half for both results); 9: as 8 without the barrier; 10: as 8
// without neg (an ADD of q's low half); 11: the packed subtraction FEEDING a packed multiply back to back (the squared
// difference, as all three kernels that failed compute it): v_pk_add_f32 t, p, q neg ; v_pk_mul_f32 d, t, t; 12: as 11 on a
// pre-negated q (no modifier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

struct Sample { unsigned got0, got1, want0, want1, p0, p1, q0, q1, lane, iter; };
struct Sample3 { float got[6], want[6], p[6], a0, a1, b0; unsigned lane, iter; };   // (victims 13+: the three packed results)
__device__ Sample3 g_s3[16];
__device__ unsigned g_n3;

template <int VICTIM>
__global__ __launch_bounds__(256) void victim_kernel(const float* in, unsigned* count, Sample* samples, int iters) {
  const int tid = threadIdx.x;
  f2 p = {in[blockIdx.x * 1024 + tid], in[blockIdx.x * 1024 + tid + 256]};
  f2 q = {in[blockIdx.x * 1024 + tid + 512], in[blockIdx.x * 1024 + tid + 768]};
  unsigned wrong = 0;
  __shared__ float s_q[2048];
  for (int j = tid; j < 2048; j += 256) s_q[j] = in[(blockIdx.x * 1024 + j) % (250 * 1024)] * 3.f;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    f2 d, nq = -q;
    float e0, e1;
    asm volatile("" : "+v"(nq));
    if (VICTIM >= 4 && VICTIM <= 10) {
      const unsigned addr = (unsigned)((i * 12) & 8188);      // uniform: a broadcast read
      unsigned va = addr;
      asm volatile("" : "+v"(va));
      if (VICTIM == 4)
        asm volatile("ds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %2, %1 neg_lo:[0,1] neg_hi:[0,1]"
                     : "=&v"(d), "=&v"(q) : "v"(p), "v"(va) : "memory");
      if (VICTIM == 5)
        asm volatile("ds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %2, %1"
                     : "=&v"(d), "=&v"(q) : "v"(p), "v"(va) : "memory");
      if (VICTIM == 6)
        asm volatile("ds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\tv_pk_add_f32 %0, %2, %1 neg_lo:[0,1] neg_hi:[0,1]"
                     : "=&v"(d), "=&v"(q) : "v"(p), "v"(va) : "memory");
      if (VICTIM == 8 || VICTIM == 10) __syncthreads();
      if (VICTIM == 8 || VICTIM == 9)
        asm volatile("ds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %2, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                     : "=&v"(d), "=&v"(q) : "v"(p), "v"(va) : "memory");
      if (VICTIM == 10)
        asm volatile("ds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %2, %1 op_sel_hi:[1,0]"
                     : "=&v"(d), "=&v"(q) : "v"(p), "v"(va) : "memory");
      if (VICTIM == 8 || VICTIM == 9) {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[0]));
      } else if (VICTIM == 10) {
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[0]));
      } else if (VICTIM == 5) {
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[1]));
      } else {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[1]));
      }
    } else if (VICTIM == 23) {   // v_fma_mixhi_f16 with op_sel on an f16 source (the HIGH 16 bits of ONE register: the operand split of
      unsigned gu, wu;           // the library's matrix-core kernels), against v_fma_mixlo_f16 on the shifted register
      unsigned va = (unsigned)((i * 12) & 8180);
      asm volatile("" : "+v"(va));
      __syncthreads();
      asm volatile("ds_read2_b32 v[100:101], %2 offset1:1\n\tds_read_b32 v102, %2 offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"
                   "v_mov_b32 v104, 0\n\t"
                   "v_fma_mixhi_f16 v104, v102, v100, -v101 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                   "v_lshrrev_b32 v105, 16, v101\n\tv_mov_b32 v106, 0\n\t"
                   "v_fma_mixlo_f16 v106, v102, v100, -v105 op_sel_hi:[0,0,1]\n\t"
                   "v_lshrrev_b32 %0, 16, v104\n\tv_and_b32 %1, 0xffff, v106"
                   : "=&v"(gu), "=&v"(wu) : "v"(va) : "v100", "v101", "v102", "v104", "v105", "v106", "memory");
      d[0] = __uint_as_float(gu); d[1] = 0.f;
      e0 = __uint_as_float(wu); e1 = 0.f;
    } else if (VICTIM >= 13 && VICTIM <= 22) {
      f2 d1, d2, p2 = {p[1], p[0]}, p3 = {q[0], q[1]};
      float a0, a1, b0;
      unsigned va = (unsigned)((i * 12) & 8180);
      asm volatile("" : "+v"(va), "+v"(p2), "+v"(p3));
      __syncthreads();
#define PKSEQ(W1, MOD)                                                                                                   \
      asm volatile("ds_read2_b32 v[100:101], %6 offset1:1\n\tds_read_b32 v102, %6 offset:8\n\ts_waitcnt lgkmcnt(" W1 ")\n\t"  \
                   "v_pk_add_f32 %0, %7, v[100:101] op_sel_hi:[1,0]" MOD "\n\t"                                          \
                   "v_pk_add_f32 %1, %8, v[100:101] op_sel:[0,1]" MOD "\n\ts_waitcnt lgkmcnt(0)\n\t"                      \
                   "v_pk_add_f32 %2, %9, v[102:103] op_sel_hi:[1,0]" MOD "\n\t"                                          \
                   "v_mov_b32 %3, v100\n\tv_mov_b32 %4, v101\n\tv_mov_b32 %5, v102"                                       \
                   : "=&v"(d), "=&v"(d1), "=&v"(d2), "=&v"(a0), "=&v"(a1), "=&v"(b0)                                      \
                   : "v"(va), "v"(p), "v"(p2), "v"(p3) : "v100", "v101", "v102", "v103", "memory")
      if (VICTIM == 13) PKSEQ("1", " neg_lo:[0,1] neg_hi:[0,1]");
      if (VICTIM == 14) PKSEQ("0", " neg_lo:[0,1] neg_hi:[0,1]");
      if (VICTIM == 15) PKSEQ("1", "");
#define PKMID(MID)                                                                                                        \
      asm volatile("ds_read2_b32 v[100:101], %6 offset1:1\n\tds_read_b32 v102, %6 offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"       \
                   "v_pk_add_f32 %0, %7, v[100:101] op_sel_hi:[1,0]\n\t" MID "\n\t"                                       \
                   "v_pk_add_f32 %2, %9, v[102:103] op_sel_hi:[1,0]\n\t"                                                 \
                   "v_mov_b32 %3, v100\n\tv_mov_b32 %4, v101\n\tv_mov_b32 %5, v102"                                       \
                   : "=&v"(d), "=&v"(d1), "=&v"(d2), "=&v"(a0), "=&v"(a1), "=&v"(b0)                                      \
                   : "v"(va), "v"(p), "v"(p2), "v"(p3) : "v100", "v101", "v102", "v103", "memory")
      if (VICTIM == 17) PKMID("v_pk_mov_b32 %1, v[100:101], v[100:101] op_sel:[1,0]");      // d1 = (A.hi, A.lo)
      if (VICTIM == 18) PKMID("v_pk_mul_f32 %1, %8, v[100:101] op_sel:[0,1]");              // d1 = p2 * A.hi
      if (VICTIM == 19) PKMID("v_pk_fma_f32 %1, %8, v[100:101], %8 op_sel:[0,1,0]");        // d1 = p2 * A.hi + p2
      if (VICTIM == 20) PKMID("v_pk_add_f32 %1, v[100:101], %8 op_sel:[1,0]");              // d1 = A.hi + p2: the pair as src0
      if (VICTIM == 21) PKMID("v_pk_mul_f32 %1, v[100:101], %8 op_sel:[1,0]");              // d1 = A.hi * p2: the pair as src0
      if (VICTIM == 22) PKMID("v_pk_mul_f32 %1, v[100:101], %8 op_sel:[1,0] op_sel_hi:[0,1]");   // d1 = (A.hi p2.lo, A.lo p2.hi)
      if (VICTIM == 16)   // neg, the SAME op_sel on all three
        asm volatile("ds_read2_b32 v[100:101], %6 offset1:1\n\tds_read_b32 v102, %6 offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %7, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %1, %8, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %2, %9, v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_mov_b32 %3, v100\n\tv_mov_b32 %4, v100\n\tv_mov_b32 %5, v102"
                     : "=&v"(d), "=&v"(d1), "=&v"(d2), "=&v"(a0), "=&v"(a1), "=&v"(b0)
                     : "v"(va), "v"(p), "v"(p2), "v"(p3) : "v100", "v101", "v102", "v103", "memory");
      float r[6];
      if (VICTIM >= 17) {
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[0]) : "v"(p[0]), "v"(a0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[1]) : "v"(p[1]), "v"(a0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[4]) : "v"(p3[0]), "v"(b0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[5]) : "v"(p3[1]), "v"(b0));
        if (VICTIM == 17) { r[2] = a1; r[3] = a0; }
        if (VICTIM == 18) {
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[2]) : "v"(p2[0]), "v"(a1));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[3]) : "v"(p2[1]), "v"(a1));
        }
        if (VICTIM == 19) {
          asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(r[2]) : "v"(p2[0]), "v"(a1));
          asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(r[3]) : "v"(p2[1]), "v"(a1));
        }
        if (VICTIM == 21) {
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[2]) : "v"(a1), "v"(p2[0]));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[3]) : "v"(a1), "v"(p2[1]));
        }
        if (VICTIM == 22) {
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[2]) : "v"(a1), "v"(p2[0]));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[3]) : "v"(a0), "v"(p2[1]));
        }
        if (VICTIM == 20) {
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[2]) : "v"(a1), "v"(p2[0]));
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[3]) : "v"(a1), "v"(p2[1]));
        }
      } else if (VICTIM == 15) {
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[0]) : "v"(p[0]), "v"(a0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[1]) : "v"(p[1]), "v"(a0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[2]) : "v"(p2[0]), "v"(a1));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[3]) : "v"(p2[1]), "v"(a1));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[4]) : "v"(p3[0]), "v"(b0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[5]) : "v"(p3[1]), "v"(b0));
      } else {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[0]) : "v"(p[0]), "v"(a0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[1]) : "v"(p[1]), "v"(a0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[2]) : "v"(p2[0]), "v"(a1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[3]) : "v"(p2[1]), "v"(a1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[4]) : "v"(p3[0]), "v"(b0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[5]) : "v"(p3[1]), "v"(b0));
      }
      {
        const float gg[6] = {d[0], d[1], d1[0], d1[1], d2[0], d2[1]};
        bool bad = false;
#pragma unroll
        for (int z = 0; z < 6; ++z) bad |= __float_as_uint(gg[z]) != __float_as_uint(r[z]);
        if (bad) {
          const unsigned sl = atomicAdd(&g_n3, 1u);
          if (sl < 16) {
#pragma unroll
            for (int z = 0; z < 6; ++z) { g_s3[sl].got[z] = gg[z]; g_s3[sl].want[z] = r[z]; }
            g_s3[sl].p[0] = p[0]; g_s3[sl].p[1] = p[1]; g_s3[sl].p[2] = p2[0]; g_s3[sl].p[3] = p2[1];
            g_s3[sl].p[4] = p3[0]; g_s3[sl].p[5] = p3[1];
            g_s3[sl].a0 = a0; g_s3[sl].a1 = a1; g_s3[sl].b0 = b0; g_s3[sl].lane = tid; g_s3[sl].iter = i;
          }
        }
      }
      // fold the three pairs into the one (d, e0, e1) the common check below compares: any difference survives the xor
      const unsigned x0 = __float_as_uint(d[0]) ^ __float_as_uint(d1[0]) ^ __float_as_uint(d2[0]);
      const unsigned x1 = __float_as_uint(d[1]) ^ __float_as_uint(d1[1]) ^ __float_as_uint(d2[1]);
      const unsigned y0 = __float_as_uint(r[0]) ^ __float_as_uint(r[2]) ^ __float_as_uint(r[4]);
      const unsigned y1 = __float_as_uint(r[1]) ^ __float_as_uint(r[3]) ^ __float_as_uint(r[5]);
      d[0] = __uint_as_float(x0); d[1] = __uint_as_float(x1);
      e0 = __uint_as_float(y0); e1 = __uint_as_float(y1);
    } else if (VICTIM == 11 || VICTIM == 12) {
      f2 t;
      if (VICTIM == 11)
        asm volatile("v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %1, %1" : "=&v"(d), "=&v"(t) : "v"(p), "v"(q));
      else
        asm volatile("v_pk_add_f32 %1, %2, %3\n\tv_pk_mul_f32 %0, %1, %1" : "=&v"(d), "=&v"(t) : "v"(p), "v"(nq));
      float u0, u1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u0) : "v"(p[0]), "v"(q[0]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u1) : "v"(p[1]), "v"(q[1]));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(e0) : "v"(u0));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(e1) : "v"(u1));
    } else if (VICTIM == 0) {
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[1]));
    } else if (VICTIM == 1) {
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(p), "v"(nq));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(q[0]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(q[1]));
    } else if (VICTIM == 2) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(nq[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(nq[1]));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(p), "v"(q));
      asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e0) : "v"(p[0]), "v"(nq[0]));
      asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e1) : "v"(p[1]), "v"(nq[1]));
    }
    const unsigned g0 = __float_as_uint(d[0]), g1 = __float_as_uint(d[1]);
    const unsigned w0 = __float_as_uint(e0), w1 = __float_as_uint(e1);
    if (g0 != w0 || g1 != w1) {
      if (wrong == 0) {
        const unsigned slot = atomicAdd(count + 1, 1u);
        if (slot < 32)
          samples[slot] = Sample{g0, g1, w0, w1, __float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(q[0]),
                                 __float_as_uint(q[1]), (unsigned)(tid & 63), (unsigned)i};
      }
      ++wrong;
    }
    p[0] += 0.0078125f; p[1] -= 0.015625f;     // new operands every iteration
    q[0] = q[0] * 1.0009765625f + 0.25f; q[1] = q[1] * 0.99951171875f - 0.125f;
    if (VICTIM < 4 && fabsf(q[0]) > 1e6f) q[0] = 1.f;
  }
  if (wrong) atomicAdd(count, wrong);
}

// the neighbour: NR independent fma chains per lane + MF dense matrix instructions per pass over them (zero data: the values
// do not matter).  MF = 0: chains only; NR = 0: matrix instructions only
template <int NR, int MF>
__global__ __launch_bounds__(512) void neighbour_kernel(float* out, int iters) {
  float r[NR > 0 ? NR : 1];
  const float b = 1.0001f, c = 0.5f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    r[i] = threadIdx.x * 1e-3f + i;
    asm volatile("" : "+v"(r[i]));
  }
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc = {};
  half8 h = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = __builtin_fmaf(r[i], b, c);
#pragma unroll
    for (int u = 0; u < MF; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, acc, 0, 0, 0);
  }
  float a = acc[0] + acc[7];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    asm volatile("" : "+v"(r[i]));
    a += r[i];
  }
  if (a == 12345.f) out[blockIdx.x] = a;
}

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main() {
  const int WG = 250;
  std::vector<float> in((size_t)WG * 1024);
  srand(7);
  for (auto& v : in) v = (rand() % 20001 - 10000) * 1e-4f;
  float *din, *dout;
  unsigned* dcount;
  Sample* dsamp;
  CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&dcount, 8)); CK(hipMalloc(&dsamp, 32 * sizeof(Sample)));
  CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int iters = 20000;    // packed instructions per lane and launch
  auto trial = [&](int victim, int nb, const char* name) -> int {
    unsigned long long wrong = 0, launches_wrong = 0;
    const int reps = victim >= 13 ? 60 : 20;
    Sample first[32];
    unsigned nsamp = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(dcount, 0, 8, sa));
      CK(hipStreamSynchronize(sa));
      if (nb == 1) hipLaunchKernelGGL((neighbour_kernel<48, 4>), dim3(256), dim3(512), 0, sb, dout, 1500);
      if (nb == 2) hipLaunchKernelGGL((neighbour_kernel<48, 0>), dim3(256), dim3(512), 0, sb, dout, 2500);
      if (nb == 3) hipLaunchKernelGGL((neighbour_kernel<0, 4>), dim3(256), dim3(512), 0, sb, dout, 4000);
      if (nb == 4) hipLaunchKernelGGL((neighbour_kernel<48, 1>), dim3(256), dim3(512), 0, sb, dout, 2000);
      if (victim == 0) hipLaunchKernelGGL(victim_kernel<0>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 1) hipLaunchKernelGGL(victim_kernel<1>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 2) hipLaunchKernelGGL(victim_kernel<2>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 3) hipLaunchKernelGGL(victim_kernel<3>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 4) hipLaunchKernelGGL(victim_kernel<4>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 5) hipLaunchKernelGGL(victim_kernel<5>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 6) hipLaunchKernelGGL(victim_kernel<6>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 11) hipLaunchKernelGGL(victim_kernel<11>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 12) hipLaunchKernelGGL(victim_kernel<12>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 13) hipLaunchKernelGGL(victim_kernel<13>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 14) hipLaunchKernelGGL(victim_kernel<14>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 15) hipLaunchKernelGGL(victim_kernel<15>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 16) hipLaunchKernelGGL(victim_kernel<16>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 17) hipLaunchKernelGGL(victim_kernel<17>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 18) hipLaunchKernelGGL(victim_kernel<18>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 19) hipLaunchKernelGGL(victim_kernel<19>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 20) hipLaunchKernelGGL(victim_kernel<20>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 23) hipLaunchKernelGGL(victim_kernel<23>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 21) hipLaunchKernelGGL(victim_kernel<21>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 22) hipLaunchKernelGGL(victim_kernel<22>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 8) hipLaunchKernelGGL(victim_kernel<8>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 9) hipLaunchKernelGGL(victim_kernel<9>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      if (victim == 10) hipLaunchKernelGGL(victim_kernel<10>, dim3(WG), dim3(256), 0, sa, din, dcount, dsamp, iters);
      CK(hipStreamSynchronize(sa));
      CK(hipStreamSynchronize(sb));
      unsigned c[2];
      CK(hipMemcpy(c, dcount, 8, hipMemcpyDeviceToHost));
      wrong += c[0];
      launches_wrong += c[0] > 0;
      if (c[0] && !nsamp) {
        nsamp = c[1] < 32 ? c[1] : 32;
        CK(hipMemcpy(first, dsamp, nsamp * sizeof(Sample), hipMemcpyDeviceToHost));
      }
    }
    const double total = (double)reps * WG * 256 * iters;   // (victims 13+: three packed instructions each)
    printf("%-58s %2d launches: %llu with a wrong value, %llu wrong lane-results of %.1e (%.1e)\n", name, reps, launches_wrong,
           wrong, total, (double)wrong / total);
    fflush(stdout);
    if (victim >= 13 && victim <= 22) {
      static Sample3 h3[16];
      unsigned n3 = 0;
      CK(hipMemcpyFromSymbol(&n3, HIP_SYMBOL(g_n3), 4));
      CK(hipMemcpyFromSymbol(h3, HIP_SYMBOL(g_s3), sizeof(h3)));
      for (unsigned i = 0; i < n3 && i < 16; i += 5) {
        const Sample3& t = h3[i];
        printf("    thread %3u iter %5u: A = (%.7g, %.7g) B0 = %.7g\n", t.lane, t.iter, t.a0, t.a1, t.b0);
        const char* nm[3] = {"d0 = p  - A.lo (op_sel_hi:[1,0])", "d1 = p2 - A.hi (op_sel:[0,1])  ", "d2 = p3 - B.lo (op_sel_hi:[1,0])"};
        for (int z = 0; z < 3; ++z)
          printf("      %s: p = (%.7g, %.7g) got (%.7g, %.7g) want (%.7g, %.7g)%s\n", nm[z], t.p[2 * z], t.p[2 * z + 1], t.got[2 * z],
                 t.got[2 * z + 1], t.want[2 * z], t.want[2 * z + 1],
                 (t.got[2 * z] != t.want[2 * z] || t.got[2 * z + 1] != t.want[2 * z + 1]) ? "   <-- WRONG" : "");
      }
      n3 = 0;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_n3), &n3, 4));
      nsamp = 0;
    }
    for (unsigned i = 0; i < nsamp && i < 6; ++i) {
      const Sample& s = first[i];
      float f[8];
      memcpy(f, &s, 32);
      printf("    lane %2u iter %5u: p = (%.9g, %.9g) q = (%.9g, %.9g) -> got (%.9g, %.9g) want (%.9g, %.9g)%s%s\n", s.lane, s.iter,
             f[4], f[5], f[6], f[7], f[0], f[1], f[2], f[3], s.got0 != s.want0 ? " [lo]" : "", s.got1 != s.want1 ? " [hi]" : "");
    }
    return 0;
  };
  if (!getenv("PK_ALL")) {
    trial(13, 0, "sampler's sequence (LDS read in flight): alone");
    trial(13, 2, "sampler's sequence: beside 48 fma chains");
    trial(13, 1, "sampler's sequence: beside chains + 4 MFMA");
    trial(14, 1, "sampler's sequence, all LDS landed first: beside chains + 4 MFMA");
    trial(15, 1, "sampler's sequence, no neg: beside chains + 4 MFMA");
    trial(16, 1, "all LDS landed, neg, ONE op_sel on all three: beside chains + 4 MFMA");
    trial(17, 1, "middle = v_pk_mov_b32 op_sel:[1,0] (swap): beside chains + 4 MFMA");
    trial(18, 1, "middle = v_pk_mul_f32 op_sel:[0,1]: beside chains + 4 MFMA");
    trial(19, 1, "middle = v_pk_fma_f32 op_sel:[0,1,0]: beside chains + 4 MFMA");
    trial(20, 1, "middle = v_pk_add_f32 op_sel:[1,0] (pair as src0): beside chains + 4 MFMA");
    trial(21, 4, "middle = v_pk_mul_f32 op_sel:[1,0] (pair as src0): beside chains + 1 MFMA");
    trial(22, 4, "middle = v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (pair as src0): beside chains + 1 MFMA");
    trial(20, 4, "middle = v_pk_add_f32 op_sel:[1,0] (pair as src0): beside chains + 1 MFMA");
    trial(17, 4, "middle = v_pk_mov_b32 op_sel:[1,0] (swap): beside chains + 1 MFMA");
    trial(16, 4, "neg, op_sel_hi:[1,0] on all three: beside chains + 1 MFMA");
    trial(23, 1, "v_fma_mixhi_f16 op_sel:[0,0,1] (f16 from the high 16 bits): beside chains + 4 MFMA");
    trial(23, 4, "v_fma_mixhi_f16 op_sel:[0,0,1]: beside chains + 1 MFMA");
    trial(13, 3, "sampler's sequence: beside matrix instructions only");
    trial(13, 4, "sampler's sequence: beside chains + 1 MFMA");
    return 0;
  }
  trial(0, 0, "v_pk_add_f32 neg: alone");
  trial(0, 2, "v_pk_add_f32 neg: beside 48 fma chains");
  trial(0, 3, "v_pk_add_f32 neg: beside matrix instructions only");
  trial(0, 1, "v_pk_add_f32 neg: beside 48 fma chains + 4 MFMA f16");
  trial(0, 4, "v_pk_add_f32 neg: beside 48 fma chains + 1 MFMA f16");
  trial(1, 1, "v_pk_add_f32 on a pre-negated q: beside chains + 4 MFMA");
  trial(2, 1, "v_pk_mul_f32 neg: beside chains + 4 MFMA");
  trial(3, 1, "v_pk_fma_f32 neg (addend): beside chains + 4 MFMA");
  trial(4, 0, "LDS operand, v_pk_add_f32 neg: alone");
  trial(4, 2, "LDS operand, v_pk_add_f32 neg: beside 48 fma chains");
  trial(4, 1, "LDS operand, v_pk_add_f32 neg: beside chains + 4 MFMA");
  trial(5, 1, "LDS operand, v_pk_add_f32 (no modifier): beside chains + 4 MFMA");
  trial(6, 1, "LDS operand, s_nop 7, v_pk_add_f32 neg: beside chains + 4 MFMA");
  trial(8, 0, "barrier, LDS operand, op_sel + neg: alone");
  trial(8, 2, "barrier, LDS operand, op_sel + neg: beside 48 fma chains");
  trial(8, 1, "barrier, LDS operand, op_sel + neg: beside chains + 4 MFMA");
  trial(9, 1, "(no barrier) LDS operand, op_sel + neg: beside chains + 4 MFMA");
  trial(10, 1, "barrier, LDS operand, op_sel, NO neg: beside chains + 4 MFMA");
  trial(11, 0, "pk sub (neg) -> pk mul, back to back: alone");
  trial(11, 2, "pk sub (neg) -> pk mul: beside 48 fma chains");
  trial(11, 3, "pk sub (neg) -> pk mul: beside matrix instructions only");
  trial(11, 1, "pk sub (neg) -> pk mul: beside chains + 4 MFMA");
  trial(12, 1, "pk add of a pre-negated q -> pk mul: beside chains + 4 MFMA");
  trial(0, 0, "v_pk_add_f32 neg: alone again");
  return 0;
}
