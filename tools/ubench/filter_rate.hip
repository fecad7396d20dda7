// filter_rate.hip -- what does ONE SIMD sustain on the primary sphere filter's instruction mix?
// (DESIGN.md "Which roofline bounds the frame kernels": the issue-cost model behind `issue_frac`)
//
//   mode 0: 12 v_pk_fma_f32 (one SGPR-pair operand each, op_sel broadcasts) + 4 v_max3_f32 |.|
//           per block of 4 spheres x 2 pixels, constants resident in SGPRs -- no loads in the loop
//   mode 1: the same body fed by s_load_dwordx16 from a 160 KB table (10,000 spheres), one block
//           per load, waits placed by the compiler
//   mode 2: v_pk_fma_f32 only (12 per block, VGPR operands only)
//   mode 3: mode 0 with the kernel's loop shape: two bodies (8 spheres) per flag check
// 256-thread workgroups, enough of them to keep 7-8 waves per SIMD resident.
//   hipcc --offload-arch=gfx950 -O3 -o filter_rate filter_rate.hip && ./filter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { v2f xw, yz; };

__device__ __forceinline__ void body(const Rec (&s)[4], v2f dx, v2f dy, v2f dz, float &m) {
  v2f b0, b1, b2, b3;
  asm volatile(
      "v_pk_fma_f32 %0, %[s0a], %[x], %[s0a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1a], %[x], %[s1a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2a], %[x], %[s2a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3a], %[x], %[s3a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %[s0b], %[y], %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1b], %[y], %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2b], %[y], %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3b], %[y], %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %[s0b], %[z], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1b], %[z], %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2b], %[z], %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3b], %[z], %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "s_nop 0"
      : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
      : [x] "v"(dx), [y] "v"(dy), [z] "v"(dz), [s0a] "s"(s[0].xw), [s0b] "s"(s[0].yz),
        [s1a] "s"(s[1].xw), [s1b] "s"(s[1].yz), [s2a] "s"(s[2].xw), [s2b] "s"(s[2].yz),
        [s3a] "s"(s[3].xw), [s3b] "s"(s[3].yz));
  asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|\n\t"
               "v_max3_f32 %0, %0, |%5|, |%6|\n\tv_max3_f32 %0, %0, |%7|, |%8|"
               : "+v"(m)
               : "v"(b0.x), "v"(b0.y), "v"(b1.x), "v"(b1.y), "v"(b2.x), "v"(b2.y), "v"(b3.x), "v"(b3.y));
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const Rec *__restrict__ tab, int n, int iters, float *out,
                                         float fx, float fy, float fz) {
  v2f dx = {fx + threadIdx.x * 1e-7f, fx * 0.5f}, dy = {fy, fy * 2.f}, dz = {fz, fz * 0.25f};
  float acc = 0.f;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  typedef const u4 __attribute__((address_space(4))) *ConstPtr;
  Rec r0[4];
  for (int i = 0; i < 4; ++i) {
    const ConstPtr src = (ConstPtr)(uintptr_t)(tab + i);
    *reinterpret_cast<u4 *>(&r0[i]) = *src;
  }
  Rec r1[4];
  for (int i = 0; i < 4; ++i) {
    const ConstPtr src = (ConstPtr)(uintptr_t)(tab + 4 + i);
    *reinterpret_cast<u4 *>(&r1[i]) = *src;
  }
  for (int it = 0; it < iters; ++it) {
    float m = 0.f;
    for (int kk = 0; kk < n; kk += 4) {
      if (MODE == 0) {
        body(r0, dx, dy, dz, m);
      } else if (MODE == 3) { // the kernel's shape: 8 spheres per flag check, m reset per check
        m = 0.f;
        body(r0, dx, dy, dz, m);
        body(r1, dx, dy, dz, m);
        kk += 4;
      } else if (MODE == 1) {
        Rec r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const ConstPtr src = (ConstPtr)(uintptr_t)(tab + kk + i);
          *reinterpret_cast<u4 *>(&r[i]) = *src;
        }
        body(r, dx, dy, dz, m);
      } else {
        v2f a = dx, b = dy, c = dz, d = dx;
        asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\t"
                     "v_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                     "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\t"
                     "v_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                     "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\t"
                     "v_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(dy), "v"(dz));
        m += a.x + b.x + c.x + d.x;
      }
      if (__builtin_amdgcn_ballot_w64(m >= 1e30f)) acc += m; // never taken; keeps m live
    }
    acc += m;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE> int run(const char *name, const Rec *d_tab, float *d_out, int n, int iters, int blocks, double instr_per_block) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, 1, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, iters, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_blocks = (double)blocks * 4 * iters * (n / 4.0);
  const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / wave_blocks; // SIMD-cycles per 4-sphere block at 2.4 GHz
  printf("%-44s %8.3f ms  %6.1f SIMD-cycles per block of 4 spheres x 128 rays (%.0f VALU instr)  = %.2f per sphere\n",
         name, ms, cyc, instr_per_block, cyc / 4);
  return 0;
}

int main() {
  const int n = 10000;
  std::vector<Rec> h(n + 8);
  for (int i = 0; i < n + 8; ++i) { h[i].xw = v2f{0.01f * (i % 97), 0.f}; h[i].yz = v2f{0.02f, -0.03f}; }
  Rec *d_tab; float *d_out;
  CHECK(hipMalloc(&d_tab, h.size() * sizeof(Rec)));
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMemcpy(d_tab, h.data(), h.size() * sizeof(Rec), hipMemcpyHostToDevice));
  const int blocks = 256 * 8 * 4; // 8 workgroups per CU resident, 4 rounds
  run<2>("mode2: 12 v_pk_fma_f32 (VGPR operands)", d_tab, d_out, n, 4, blocks, 12);
  run<0>("mode0: filter body, constants in SGPRs", d_tab, d_out, n, 4, blocks, 16);
  run<1>("mode1: filter body + s_load per block", d_tab, d_out, n, 4, blocks, 16);
  run<3>("mode3: 2 bodies (8 spheres) per flag check", d_tab, d_out, n, 4, blocks, 16);
  return 0;
}
