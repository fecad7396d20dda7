// Microbenchmark: what wave64 VALU issue rate does gfx950 actually sustain for the instruction
// mix of k_render's inner loops (v_mul_f32 / v_add_f32 / v_sub_f32, no FMA, SGPR operands)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: independent mul/add on VGPRs only (8 accumulators)
// MODE 1: the primary sphere test body on SGPR operands, records from a uniform pointer (s_load), no branch
// MODE 2: MODE 1 + the max/cmp/branch filter (never taken)
// MODE 3: same as 2 but table in LDS (ds_read_b128 broadcast)
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ tab, int n, int iters, float *out,
                                         float dx, float dy, float dz) {
  __shared__ float4 lds[2048];
  float acc = 0.f;
  float x = dx + threadIdx.x * 1e-6f, y = dy, z = dz;
  if (MODE == 3) {
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = tab[i];
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      float a0 = x, a1 = y, a2 = z, a3 = x + 1.f, a4 = y + 1.f, a5 = z + 1.f, a6 = x + 2.f, a7 = y + 2.f;
      for (int k = 0; k < n; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = a0 * x; a1 = a1 + y; a2 = a2 * z; a3 = a3 + x;
          a4 = a4 * y; a5 = a5 + z; a6 = a6 * x; a7 = a7 + y;
        }
      }
      acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
      for (int k = 0; k < n; k += 4) {
        float4 s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = (MODE == 3) ? lds[(k + u) & 2047] : tab[k + u];
        float q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float b = (s[u].x * x + s[u].y * y) + s[u].z * z;
          q[u] = b * b - s[u].w;
        }
        if (MODE >= 2) {
          float m = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
          if (__builtin_amdgcn_ballot_w64(!(m < 0.f))) acc += sqrtf(q[0]) + sqrtf(q[1]) + sqrtf(q[2]) + sqrtf(q[3]);
        } else {
          acc += q[0]; acc += q[1]; acc += q[2]; acc += q[3];
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 4: packed fp32 (v_pk_mul_f32 / v_pk_add_f32), 8 independent float2 accumulators
__global__ void __launch_bounds__(256) kpk(int n, int iters, float *out, float dx, float dy) {
  float acc = 0.f;
  v2f x = {dx + threadIdx.x * 1e-6f, dx * 0.5f}, y = {dy, dy * 2.f};
  for (int it = 0; it < iters; ++it) {
    v2f a0 = x, a1 = y, a2 = x + 1.f, a3 = y + 1.f, a4 = x + 2.f, a5 = y + 2.f, a6 = x + 3.f, a7 = y + 3.f;
    for (int k = 0; k < n; k += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = a0 * x; a1 = a1 + y; a2 = a2 * y; a3 = a3 + x;
        a4 = a4 * y; a5 = a5 + x; a6 = a6 * x; a7 = a7 + y;
      }
    }
    v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    acc += t.x + t.y;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int run_pk(float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, n, 1, d_out, 0.3f, 0.5f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, n, iters, d_out, 0.3f, 0.5f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instr = (double)blocks * 4 * iters * (n / 4.0) * 32;
  double per = wave_instr / 1024.0 / (ms * 1e3);
  printf("%-28s blocks %6d  %8.3f ms  %6.1f PK wave-instr/us/SIMD  -> %.2f cycles/instr @2.4GHz (%.1f Tlane-op/s, 2 ops per pk lane)\n",
         "mode4 v_pk mul/add x32", blocks, ms, per, 2400.0 / per, wave_instr * 128 / (ms * 1e-3) / 1e12);
  return 0;
}

// MODE 5: the primary sphere test for 2 pixels per lane, 4 spheres per block, hand-written:
// packed fp32 with the sphere constants read straight from SGPR pairs through op_sel (no moves),
// the four independent chains interleaved so no v_pk result is consumed by the next instruction.
__global__ void __launch_bounds__(256) kasm(const float4 *__restrict__ tab, int n, int iters, float *out,
                                            float dx, float dy, float dz) {
  float acc = 0.f;
  v2f x = {dx + threadIdx.x * 1e-6f, dx * 0.5f}, y = {dy, dy * 2.f}, z = {dz, dz * 0.25f};
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < n; k += 4) {
      float4 s0 = tab[k], s1 = tab[k + 1], s2 = tab[k + 2], s3 = tab[k + 3];
      v2f q0, q1, q2, q3, t0, t1, t2, t3;
      float m;
      asm volatile(
          "v_pk_mul_f32 %0, %[s0xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %1, %[s1xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %2, %[s2xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %3, %[s3xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %4, %[s0xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %5, %[s1xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %6, %[s2xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %7, %[s3xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_add_f32 %0, %0, %4\n\t"
          "v_pk_add_f32 %1, %1, %5\n\t"
          "v_pk_add_f32 %2, %2, %6\n\t"
          "v_pk_add_f32 %3, %3, %7\n\t"
          "v_pk_mul_f32 %4, %[s0zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %5, %[s1zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %6, %[s2zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %7, %[s3zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_add_f32 %0, %0, %4\n\t"
          "v_pk_add_f32 %1, %1, %5\n\t"
          "v_pk_add_f32 %2, %2, %6\n\t"
          "v_pk_add_f32 %3, %3, %7\n\t"
          "v_pk_mul_f32 %4, %0, %0\n\t"
          "v_pk_mul_f32 %5, %1, %1\n\t"
          "v_pk_mul_f32 %6, %2, %2\n\t"
          "v_pk_mul_f32 %7, %3, %3\n\t"
          "v_pk_add_f32 %4, %4, %[s0zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %5, %5, %[s1zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %6, %6, %[s2zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %7, %7, %[s3zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
          : [x] "v"(x), [y] "v"(y), [z] "v"(z),
            [s0xy] "s"(*(const v2f *)&s0.x), [s0zw] "s"(*(const v2f *)&s0.z),
            [s1xy] "s"(*(const v2f *)&s1.x), [s1zw] "s"(*(const v2f *)&s1.z),
            [s2xy] "s"(*(const v2f *)&s2.x), [s2zw] "s"(*(const v2f *)&s2.z),
            [s3xy] "s"(*(const v2f *)&s3.x), [s3zw] "s"(*(const v2f *)&s3.z));
      m = fmaxf(fmaxf(fmaxf(q0.x, q0.y), fmaxf(q1.x, q1.y)), fmaxf(fmaxf(q2.x, q2.y), fmaxf(q3.x, q3.y)));
      if (__builtin_amdgcn_ballot_w64(!(m < 0.f)))
        acc += sqrtf(q0.x) + sqrtf(q1.y) + sqrtf(q2.x) + sqrtf(q3.y) + t0.x + t1.y + t2.x + t3.y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int run_asm(const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kasm, dim3(blocks), dim3(256), 0, 0, d_tab, n, 1, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kasm, dim3(blocks), dim3(256), 0, 0, d_tab, n, iters, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double tests = (double)blocks * 4 * iters * n * 128.0;  // pixel-tests
  printf("%-28s blocks %6d  %8.3f ms  %.1f Gtests/s (pixel x sphere)\n", "mode5 asm pk smem px2", blocks, ms, tests / (ms * 1e-3) / 1e9);
  return 0;
}

// MODE 6/7: does a v_pk op with two VGPR-pair sources pay for VGPR bank conflicts?  16 independent
// v_pk_mul_f32 per block with explicit registers: sources in the SAME banks (v[8:9] x v[12:13],
// both pairs start at bank 0) versus in DIFFERENT banks (v[8:9] x v[14:15]).

template <int CONFLICT>
__global__ void __launch_bounds__(256) kbank(int n, float *out) {
  float acc = 0.f;
  for (int k = 0; k < n; ++k) {
    if (CONFLICT) {
      asm volatile(
          "v_pk_mul_f32 v[16:17], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[16:17], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[12:13]\n\t"
          ::: "v8","v9","v12","v13","v14","v15","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v44","v45");
    } else {
      asm volatile(
          "v_pk_mul_f32 v[16:17], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[16:17], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[14:15]\n\t"
          ::: "v8","v9","v12","v13","v14","v15","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v44","v45");
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int CONFLICT> int run_bank(float *d_out, int blocks) {
  const int n = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kbank<CONFLICT>, dim3(blocks), dim3(256), 0, 0, 16, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kbank<CONFLICT>, dim3(blocks), dim3(256), 0, 0, n, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instr = (double)blocks * 4 * n * 16;
  double per = wave_instr / 1024.0 / (ms * 1e3);
  printf("mode%d v_pk src banks %-9s blocks %6d  %8.3f ms  -> %.2f cycles/pk-instr @2.4GHz\n", 6 + CONFLICT,
         CONFLICT ? "same" : "different", blocks, ms, 2400.0 / per);
  return 0;
}

// MODE 8: the kernel's shadow-ray pair body (32 v_pk per 4 spheres, 1 ray per lane) in isolation
struct PairG { v2f x, y, z, r; };
__device__ __forceinline__ void pair2_any_pk(const PairG (&R)[2], v2f oxy, v2f oz_, v2f Lxy, v2f Lz_, v2f (&b)[2],
                         v2f (&q)[2]) {
  // Scheduling rule measured in tools/ubench/valu_rate.hip (modes 8/10): a v_pk result must not
  // be consumed within the next 3 instructions of the same wave (other waves do not fill the
  // gap): 227 -> 148 cycles per block.  Four chains are kept in flight: dot(oc,L) and dot(oc,oc)
  // of record 0 and of record 1.
  v2f ax, ay, az, bx, by, bz, t0, t1, u0, u1;
  asm("v_pk_add_f32 %[ax], %[oxy], %[r0x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bx], %[oxy], %[r1x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay], %[oxy], %[r0y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[by], %[oxy], %[r1y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az], %[oz], %[r0z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bz], %[oz], %[r1z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[b0], %[ax], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b1], %[bx], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[q0], %[ax], %[ax]\n\t"
      "v_pk_mul_f32 %[q1], %[bx], %[bx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[u0], %[ay], %[ay]\n\t"
      "v_pk_mul_f32 %[u1], %[by], %[by]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[u0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[u1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[u0], %[az], %[az]\n\t"
      "v_pk_mul_f32 %[u1], %[bz], %[bz]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[u0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[u1]\n\t"
      "v_pk_mul_f32 %[t0], %[b0], %[b0]\n\t"
      "v_pk_mul_f32 %[t1], %[b1], %[b1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[r0r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[r1r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1\n\t"
      "v_pk_add_f32 %[q0], %[t0], %[q0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[t1], %[q1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [q0] "=&v"(q[0]), [q1] "=&v"(q[1]), [ax] "=&v"(ax),
        [ay] "=&v"(ay), [az] "=&v"(az), [bx] "=&v"(bx), [by] "=&v"(by), [bz] "=&v"(bz),
        [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1)
      : [oxy] "v"(oxy), [oz] "v"(oz_), [Lxy] "v"(Lxy), [Lz] "v"(Lz_), [r0x] "s"(R[0].x),
        [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0r] "s"(R[0].r), [r1x] "s"(R[1].x),
        [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1r] "s"(R[1].r));
}

__device__ __forceinline__ void pair2_any_pk_splat(const PairG (&R)[2], v2f ox2, v2f oy2, v2f oz2, v2f Lx2, v2f Ly2, v2f Lz2,
                                                    v2f (&b)[2], v2f (&q)[2]) {
  v2f ax, ay, az, bx, by, bz, t0, t1;
  asm("v_pk_add_f32 %[ax], %[r0x], %[ox] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[bx], %[r1x], %[ox] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[ay], %[r0y], %[oy] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[by], %[r1y], %[oy] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[az], %[r0z], %[oz] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[bz], %[r1z], %[oz] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b0], %[ax], %[Lx]\n\t"
      "v_pk_mul_f32 %[b1], %[bx], %[Lx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[Ly]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[Ly]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[Lz]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[Lz]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_mul_f32 %[q0], %[ax], %[ax]\n\t"
      "v_pk_mul_f32 %[q1], %[bx], %[bx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[ay]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[by]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[t0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[t1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[az]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[bz]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[t0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[t1]\n\t"
      "v_pk_add_f32 %[q0], %[r0r], %[q0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_add_f32 %[q1], %[r1r], %[q1] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t0], %[b0], %[b0]\n\t"
      "v_pk_mul_f32 %[t1], %[b1], %[b1]\n\t"
      "v_pk_add_f32 %[q0], %[t0], %[q0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[t1], %[q1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [q0] "=&v"(q[0]), [q1] "=&v"(q[1]), [ax] "=&v"(ax),
        [ay] "=&v"(ay), [az] "=&v"(az), [bx] "=&v"(bx), [by] "=&v"(by), [bz] "=&v"(bz),
        [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [ox] "v"(ox2), [oy] "v"(oy2), [oz] "v"(oz2), [Lx] "v"(Lx2), [Ly] "v"(Ly2), [Lz] "v"(Lz2), [r0x] "s"(R[0].x),
        [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0r] "s"(R[0].r), [r1x] "s"(R[1].x),
        [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1r] "s"(R[1].r));
}

__device__ __forceinline__ void pairN_any_pk(const PairG (&R)[4], v2f oxy, v2f oz_, v2f Lxy, v2f Lz_, v2f (&b)[4], v2f (&q)[4]) {
  v2f ax[4], ay[4], az[4], t[4];
  asm("v_pk_add_f32 %[ax0], %[oxy], %[r0x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ax1], %[oxy], %[r1x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ax2], %[oxy], %[r2x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ax3], %[oxy], %[r3x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay0], %[oxy], %[r0y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay1], %[oxy], %[r1y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay2], %[oxy], %[r2y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay3], %[oxy], %[r3y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az0], %[oz], %[r0z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az1], %[oz], %[r1z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az2], %[oz], %[r2z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az3], %[oz], %[r3z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[b0], %[ax0], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b1], %[ax1], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b2], %[ax2], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b3], %[ax3], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t0], %[ay0], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t1], %[ay1], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t2], %[ay2], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t3], %[ay3], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[b2], %[b2], %[t2]\n\t"
      "v_pk_add_f32 %[b3], %[b3], %[t3]\n\t"
      "v_pk_mul_f32 %[t0], %[az0], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t1], %[az1], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t2], %[az2], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t3], %[az3], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[b2], %[b2], %[t2]\n\t"
      "v_pk_add_f32 %[b3], %[b3], %[t3]\n\t"
      "v_pk_mul_f32 %[q0], %[ax0], %[ax0]\n\t"
      "v_pk_mul_f32 %[q1], %[ax1], %[ax1]\n\t"
      "v_pk_mul_f32 %[q2], %[ax2], %[ax2]\n\t"
      "v_pk_mul_f32 %[q3], %[ax3], %[ax3]\n\t"
      "v_pk_mul_f32 %[t0], %[ay0], %[ay0]\n\t"
      "v_pk_mul_f32 %[t1], %[ay1], %[ay1]\n\t"
      "v_pk_mul_f32 %[t2], %[ay2], %[ay2]\n\t"
      "v_pk_mul_f32 %[t3], %[ay3], %[ay3]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[t0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[t1]\n\t"
      "v_pk_add_f32 %[q2], %[q2], %[t2]\n\t"
      "v_pk_add_f32 %[q3], %[q3], %[t3]\n\t"
      "v_pk_mul_f32 %[t0], %[az0], %[az0]\n\t"
      "v_pk_mul_f32 %[t1], %[az1], %[az1]\n\t"
      "v_pk_mul_f32 %[t2], %[az2], %[az2]\n\t"
      "v_pk_mul_f32 %[t3], %[az3], %[az3]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[t0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[t1]\n\t"
      "v_pk_add_f32 %[q2], %[q2], %[t2]\n\t"
      "v_pk_add_f32 %[q3], %[q3], %[t3]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[r0r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[r1r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q2], %[q2], %[r2r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q3], %[q3], %[r3r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[t0], %[b0], %[b0]\n\t"
      "v_pk_mul_f32 %[t1], %[b1], %[b1]\n\t"
      "v_pk_mul_f32 %[t2], %[b2], %[b2]\n\t"
      "v_pk_mul_f32 %[t3], %[b3], %[b3]\n\t"
      "v_pk_add_f32 %[q0], %[t0], %[q0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[t1], %[q1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q2], %[t2], %[q2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q3], %[t3], %[q3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [b0] "=&v"(b[0]), [q0] "=&v"(q[0]), [ax0] "=&v"(ax[0]), [ay0] "=&v"(ay[0]), [az0] "=&v"(az[0]), [t0] "=&v"(t[0]), [b1] "=&v"(b[1]), [q1] "=&v"(q[1]), [ax1] "=&v"(ax[1]), [ay1] "=&v"(ay[1]), [az1] "=&v"(az[1]), [t1] "=&v"(t[1]), [b2] "=&v"(b[2]), [q2] "=&v"(q[2]), [ax2] "=&v"(ax[2]), [ay2] "=&v"(ay[2]), [az2] "=&v"(az[2]), [t2] "=&v"(t[2]), [b3] "=&v"(b[3]), [q3] "=&v"(q[3]), [ax3] "=&v"(ax[3]), [ay3] "=&v"(ay[3]), [az3] "=&v"(az[3]), [t3] "=&v"(t[3])
      : [oxy] "v"(oxy), [oz] "v"(oz_), [Lxy] "v"(Lxy), [Lz] "v"(Lz_), [r0x] "s"(R[0].x), [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0r] "s"(R[0].r), [r1x] "s"(R[1].x), [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1r] "s"(R[1].r), [r2x] "s"(R[2].x), [r2y] "s"(R[2].y), [r2z] "s"(R[2].z), [r2r] "s"(R[2].r), [r3x] "s"(R[3].x), [r3y] "s"(R[3].y), [r3z] "s"(R[3].z), [r3r] "s"(R[3].r));
}

__global__ void __launch_bounds__(256) kshadow4(const PairG *__restrict__ tab, int n_rec, int iters, float *out,
                                                float ox, float oy, float oz) {
  float acc = 0.f;
  v2f oxy = {ox + threadIdx.x * 1e-6f, oy}, oz_ = {oz, 0.f}, Lxy = {0.6f, 0.0f}, Lz_ = {0.8f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k + 4 <= n_rec; k += 4) {
      const PairG R[4] = {tab[k], tab[k + 1], tab[k + 2], tab[k + 3]};
      v2f b[4], q[4];
      pairN_any_pk(R, oxy, oz_, Lxy, Lz_, b, q);
      int m = max(max(max(__float_as_int(q[0].x), __float_as_int(q[0].y)), __float_as_int(q[1].x)), __float_as_int(q[1].y));
      m = max(m, max(max(max(__float_as_int(q[2].x), __float_as_int(q[2].y)), __float_as_int(q[3].x)), __float_as_int(q[3].y)));
      if (__builtin_amdgcn_ballot_w64(m >= 0)) acc += sqrtf(q[0].x) + b[0].y + b[1].x + q[1].y + b[2].x + b[3].y + q[2].y + q[3].x;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int run_shadow4(const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n_rec = n / 2;
  hipLaunchKernelGGL(kshadow4, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, 1, d_out, 100.f, 200.f, 300.f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kshadow4, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, iters, d_out, 100.f, 200.f, 300.f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_blocks = (double)blocks * 4 * iters * (n_rec / 2);  // in 4-sphere units
  double per = wave_blocks / 1024.0 / (ms * 1e-3);
  printf("%-28s blocks %6d  %8.3f ms  -> %.1f cycles per 4-sphere unit @2.4GHz (8 spheres per asm, 4 chains)\n",
         "mode10 shadow body 4 chains", blocks, ms, 2.4e9 / per);
  return 0;
}

// MODE 11: two 2-record bodies per loop iteration (8 spheres per fetch / filter / branch)
__global__ void __launch_bounds__(256) kshadow2x(const PairG *__restrict__ tab, int n_rec, int iters, float *out,
                                                 float ox, float oy, float oz) {
  float acc = 0.f;
  v2f oxy = {ox + threadIdx.x * 1e-6f, oy}, oz_ = {oz, 0.f}, Lxy = {0.6f, 0.0f}, Lz_ = {0.8f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k + 4 <= n_rec; k += 4) {
      const PairG R0[2] = {tab[k], tab[k + 1]};
      const PairG R1[2] = {tab[k + 2], tab[k + 3]};
      v2f b0[2], q0[2], b1[2], q1[2];
      pair2_any_pk(R0, oxy, oz_, Lxy, Lz_, b0, q0);
      pair2_any_pk(R1, oxy, oz_, Lxy, Lz_, b1, q1);
      int m = max(max(max(__float_as_int(q0[0].x), __float_as_int(q0[0].y)), __float_as_int(q0[1].x)), __float_as_int(q0[1].y));
      m = max(m, max(max(max(__float_as_int(q1[0].x), __float_as_int(q1[0].y)), __float_as_int(q1[1].x)), __float_as_int(q1[1].y)));
      if (__builtin_amdgcn_ballot_w64(m >= 0)) acc += sqrtf(q0[0].x) + b0[0].y + b0[1].x + q0[1].y + b1[0].x + b1[1].y + q1[0].y + q1[1].x;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

// MODE 12: same, but each wave starts its sweep at its own offset, so co-resident waves do not
// find each other's lines in the scalar cache (what happens in k_render, where waves drift apart)
__global__ void __launch_bounds__(256) kshadow2x_off(const PairG *__restrict__ tab, int n_rec, int iters, float *out,
                                                 float ox, float oy, float oz) {
  float acc = 0.f;
  v2f oxy = {ox + threadIdx.x * 1e-6f, oy}, oz_ = {oz, 0.f}, Lxy = {0.6f, 0.0f}, Lz_ = {0.8f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int off = __builtin_amdgcn_readfirstlane((int)((wave_id * 2654435761u) % (unsigned)(n_rec / 4)) * 4);
    for (int kk = 0; kk + 4 <= n_rec; kk += 4) {
      int k = kk + off; if (k >= n_rec - 3) k -= (n_rec & ~3);
      const PairG R0[2] = {tab[k], tab[k + 1]};
      const PairG R1[2] = {tab[k + 2], tab[k + 3]};
      v2f b0[2], q0[2], b1[2], q1[2];
      pair2_any_pk(R0, oxy, oz_, Lxy, Lz_, b0, q0);
      pair2_any_pk(R1, oxy, oz_, Lxy, Lz_, b1, q1);
      int m = max(max(max(__float_as_int(q0[0].x), __float_as_int(q0[0].y)), __float_as_int(q0[1].x)), __float_as_int(q0[1].y));
      m = max(m, max(max(max(__float_as_int(q1[0].x), __float_as_int(q1[0].y)), __float_as_int(q1[1].x)), __float_as_int(q1[1].y)));
      if (__builtin_amdgcn_ballot_w64(m >= 0)) acc += sqrtf(q0[0].x) + b0[0].y + b0[1].x + q0[1].y + b1[0].x + b1[1].y + q1[0].y + q1[1].x;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int OFF> int run_shadow2x(const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  auto kern = OFF ? kshadow2x_off : kshadow2x;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n_rec = n / 2;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, 1, d_out, 100.f, 200.f, 300.f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, iters, d_out, 100.f, 200.f, 300.f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_blocks = (double)blocks * 4 * iters * (n_rec / 2);
  double per = wave_blocks / 1024.0 / (ms * 1e-3);
  printf("%-28s blocks %6d  %8.3f ms  -> %.1f cycles per 4-sphere unit @2.4GHz (two 2-record bodies per iteration)\n",
         OFF ? "mode12 +per-wave offsets" : "mode11 shadow 2x body/iter", blocks, ms, 2.4e9 / per);
  return 0;
}


template <int VARIANT>
__global__ void __launch_bounds__(256) kshadow(const PairG *__restrict__ tab, int n_rec, int iters, float *out,
                                               float ox, float oy, float oz) {
  float acc = 0.f;
  v2f oxy = {ox + threadIdx.x * 1e-6f, oy}, oz_ = {oz, 0.f}, Lxy = {0.6f, 0.0f}, Lz_ = {0.8f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k + 2 <= n_rec; k += 2) {
      const PairG R[2] = {tab[k], tab[k + 1]};
      v2f b[2], q[2];
      if (VARIANT == 0) pair2_any_pk(R, oxy, oz_, Lxy, Lz_, b, q);
      else pair2_any_pk_splat(R, (v2f){oxy.x, oxy.x}, (v2f){oxy.y, oxy.y}, (v2f){oz_.x, oz_.x}, (v2f){Lxy.x, Lxy.x}, (v2f){Lxy.y, Lxy.y}, (v2f){Lz_.x, Lz_.x}, b, q);
      const int m = max(max(max(__float_as_int(q[0].x), __float_as_int(q[0].y)), __float_as_int(q[1].x)), __float_as_int(q[1].y));
      if (__builtin_amdgcn_ballot_w64(m >= 0)) acc += sqrtf(q[0].x) + b[0].y + b[1].x + q[1].y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int VARIANT> int run_shadow(const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n_rec = n / 2;
  hipLaunchKernelGGL(kshadow<VARIANT>, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, 1, d_out, 100.f, 200.f, 300.f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kshadow<VARIANT>, dim3(blocks), dim3(256), 0, 0, (const PairG *)d_tab, n_rec, iters, d_out, 100.f, 200.f, 300.f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_blocks = (double)blocks * 4 * iters * (n_rec / 2);
  double per = wave_blocks / 1024.0 / (ms * 1e-3);   // blocks per SIMD per second
  printf("%-28s blocks %6d  %8.3f ms  -> %.1f cycles per 4-sphere block @2.4GHz (model 32*4.1+3*2.7 = 139)\n",
         VARIANT ? "mode9 shadow body, splat regs" : "mode8 shadow pair body smem", blocks, ms, 2.4e9 / per);
  return 0;
}

template <int MODE> int run(const char *name, int valu_per_4, const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, 1, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, iters, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double waves = (double)blocks * 4;
  double wave_instr = waves * iters * (n / 4.0) * valu_per_4;
  double per_simd_per_us = wave_instr / 1024.0 / (ms * 1e3);
  printf("%-28s blocks %6d  %8.3f ms  %6.1f VALU wave-instr/us/SIMD  -> %.2f cycles/instr @2.4GHz (%.1f Tlane-op/s) %.1f Gtests/s\n",
         name, blocks, ms, per_simd_per_us, 2400.0 / per_simd_per_us, wave_instr * 64 / (ms * 1e-3) / 1e12,
         waves * iters * (double)n * 64.0 / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  const int n = 10000;
  std::vector<float4> h(n);
  for (int i = 0; i < n; i++) h[i] = make_float4(0.01f * i, 1.f, 2.f, 1e9f);
  float4 *d_tab; float *d_out;
  CHECK(hipMalloc((void **)&d_tab, n * sizeof(float4)));
  CHECK(hipMalloc((void **)&d_out, 4));
  CHECK(hipMemcpy(d_tab, h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  for (int blocks : {2048, 8192}) {
    int iters = 16384 * 8 / blocks; if (iters < 1) iters = 1; if (iters > 64) iters = 64;
    if (run<0>("mode0 vgpr mul/add x32", 32, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<1>("mode1 sphere body smem", 32, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<2>("mode2 +filter/branch smem", 31, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<3>("mode3 +filter/branch lds", 31, d_tab, d_out, n, iters, blocks)) return 1;
    if (run_pk(d_out, n, iters, blocks)) return 1;
    if (run_asm(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_shadow<0>(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_shadow<1>(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_shadow4(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_shadow2x<0>(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_shadow2x<1>(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_bank<0>(d_out, blocks)) return 1;
    if (run_bank<1>(d_out, blocks)) return 1;
  }
  return 0;
}
