// Hardware probes behind the roofline discussion in DESIGN.md (one JSON line per measurement):
//   mfma        what the fp32 MFMA pipe sustains (TFLOP/s, implied clock) with no memory traffic: every wave
//               runs `iters` x 16 v_mfma_f32_32x32x2_f32 (1, 2 or 4 accumulators; 1 or 2 waves per SIMD);
//   coissue     does VALU / LDS work of one wave proceed under the MFMAs of its SIMD mate?
//   interleave  how many VALU instructions fit in the shadow of a 64-cycle MFMA of the SAME wave?
//               (answer on gfx950: none - fp32 MFMA time and VALU time add up)
//   stream_read the read-only HBM ceiling (what K1 is, minus its segment logic), plain and nontemporal.
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe.hip -o build/hw_probe && build/hw_probe [iters | r]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16 / NACC; ++k)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;  // keep the chain alive
}

// Co-issue probe: waves 0-3 of a workgroup (one per SIMD) run the MFMA loop, waves 4-7 (their SIMD mates)
// run `valu_iters` x 64 fp32 FMAs (4 independent chains) or, with lds != 0, ds_read_b128 + FMA.  Each role
// reports its own cycle count (s_memtime), so one sees whether VALU / LDS work of one wave proceeds at full
// rate under the other wave's MFMAs.
__global__ __launch_bounds__(512) void coissue(float* out, unsigned long long* cyc, int mfma_iters, int valu_iters,
                                               int mode, float a0) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = a0 * k;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  float s = 0.f;
  const int flags = mode >> 4;
  mode &= 15;
  if (wave >= 4 && (flags & 2)) __builtin_amdgcn_s_setprio(3);
  if (wave < 4 && (flags & 1)) {
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f;
    for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a0, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else if (wave < 4) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f;
    for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a0, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
  } else if (mode == 0) {
    float x0 = a0, x1 = a0 + 1, x2 = a0 + 2, x3 = a0 + 3;
    for (int it = 0; it < valu_iters; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        x0 = __builtin_fmaf(x0, a0, 1.f); x1 = __builtin_fmaf(x1, a0, 1.f);
        x2 = __builtin_fmaf(x2, a0, 1.f); x3 = __builtin_fmaf(x3, a0, 1.f);
      }
    s = x0 + x1 + x2 + x3;
  } else if (mode == 2) {  // scalar-ALU mate: 64 dependent s_add_u32 per iteration
    unsigned acc = (unsigned)mfma_iters;
    for (int it = 0; it < valu_iters; ++it)
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("s_add_u32 %0, %0, 3" : "+s"(acc) :: "memory");
    s = (float)acc;
  } else {
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    for (int it = 0; it < valu_iters; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) x += p[(k * 64 + it) & 0x3c0];
    s = x.x + x.y + x.z + x.w;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  if (s == 123.456f) out[0] = s;
}

// Same-wave interleave: after every MFMA of a dependent chain, KV independent v_fma_f32 (4 chains).  Shows how
// many VALU instructions fit in the shadow of one 64-cycle MFMA when they come from the SAME wave.
// Same question for instructions that do not use the vector ALU: KS scalar adds, or KS LDS reads, after every MFMA
// of the same wave.
template <int KS, bool LDSR>
__global__ __launch_bounds__(256) void interleave_other(float* out, unsigned long long* cyc, int iters, float a0) {
  __shared__ float lds[1024];
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) lds[k] = a0 * k;
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f;
  unsigned sacc = (unsigned)iters;
  float lv[KS > 0 ? KS : 1];
#pragma unroll
  for (int v = 0; v < KS; ++v) lv[v] = 0.f;
  const unsigned laddr = (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(a0));
#pragma unroll
      for (int v = 0; v < KS; ++v) {
        if (LDSR) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(lv[v]) : "v"(laddr), "n"(v * 256));
        else asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) :: "memory");
      }
    }
    if (LDSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sacc;
#pragma unroll
  for (int v = 0; v < KS; ++v) s += lv[v];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  if (s == 123.456f) out[0] = s;
}

template <int KS, bool LDSR>
static void run_interleave_other(int iters) {
  float* out;
  unsigned long long *cyc, h[8];
  hipMalloc(&out, 4);
  hipMalloc(&cyc, 64);
  interleave_other<KS, LDSR><<<256, 256>>>(out, cyc, iters, 1e-3f);
  hipDeviceSynchronize();
  interleave_other<KS, LDSR><<<256, 256>>>(out, cyc, iters, 1e-3f);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("{\"probe\": \"interleave\", \"waves_per_simd\": 1, \"%s_per_mfma\": %d, \"cycles_per_mfma_slot\": %.1f}\n",
         LDSR ? "ds_read_b32" : "s_add_u32", KS, (double)h[0] / (iters * 8.0));
  hipFree(out);
  hipFree(cyc);
}

template <int KV>
__global__ __launch_bounds__(512) void interleave(float* out, unsigned long long* cyc, int iters, float a0) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f;
  float x[4] = {a0, a0 + 1, a0 + 2, a0 + 3};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(a0));
#pragma unroll
      for (int v = 0; v < KV; ++v) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[v & 3]) : "v"(a0));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = x[0] + x[1] + x[2] + x[3];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  if (s == 123.456f) out[0] = s;
}

template <int KV>
static void run_interleave(int iters, int waves) {
  float* out;
  unsigned long long *cyc, h[8];
  hipMalloc(&out, 4);
  hipMalloc(&cyc, 64);
  interleave<KV><<<256, waves * 64>>>(out, cyc, iters, 1e-3f);
  hipDeviceSynchronize();
  interleave<KV><<<256, waves * 64>>>(out, cyc, iters, 1e-3f);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("{\"probe\": \"interleave\", \"waves_per_simd\": %d, \"valu_per_mfma\": %d, \"cycles_per_mfma_slot\": %.1f}\n", waves / 4,
         KV, (double)h[0] / (iters * 8.0));
  hipFree(out);
  hipFree(cyc);
}

static void run_coissue(int mfma_iters, int valu_iters, int mode) {
  float* out;
  unsigned long long *cyc, h[8];
  hipMalloc(&out, 4);
  hipMalloc(&cyc, 64);
  coissue<<<256, 512>>>(out, cyc, mfma_iters, valu_iters, mode, 1e-3f);
  hipDeviceSynchronize();
  coissue<<<256, 512>>>(out, cyc, mfma_iters, valu_iters, mode, 1e-3f);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double mf = mfma_iters ? (double)h[0] / (mfma_iters * 16.0) : 0.0;
  const double va = valu_iters ? (double)h[4] / (valu_iters * 64.0) : 0.0;
  printf("{\"probe\": \"coissue\", \"flags\": %d, \"mate\": \"%s\", \"mfma_iters\": %d, \"mate_iters\": %d, \"cycles_per_mfma\": %.1f, "
         "\"cycles_per_mate_instr\": %.2f}\n", mode >> 4, (mode & 15) == 0 ? "v_fma_f32" : (mode & 15) == 2 ? "s_add_u32" : "ds_read_b128+4 v_add", mfma_iters, valu_iters, mf,
         (mode & 15) == 1 ? va * 4.0 : va);
  hipFree(out);
  hipFree(cyc);
}

// Pure streaming read (what K1 is, minus its segment logic): every lane sums 16-B loads, U in flight, grid-
// stride over `n4` float4s.  Gives the read-only HBM ceiling next to the 8 TB/s spec and the copy ceiling.
template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_read(const f32x4* __restrict__ p, size_t n4, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) a += r[u];
  }
  for (; i < n4; i += stride) a += p[i];
  if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

template <int U, bool NT>
static void run_read(size_t bytes, int blocks_per_cu) {
  f32x4* p;
  float* out;
  hipMalloc(&p, bytes);
  hipMalloc(&out, 4);
  hipMemset(p, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t n4 = bytes / 16;
  stream_read<U, NT><<<256 * blocks_per_cu, 256>>>(p, n4, out);
  hipDeviceSynchronize();
  float sum = 0.f;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    stream_read<U, NT><<<256 * blocks_per_cu, 256>>>(p, n4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
  }
  printf("{\"probe\": \"stream_read\", \"GB\": %.2f, \"loads_in_flight\": %d, \"nontemporal\": %s, \"blocks_per_cu\": %d, "
         "\"ms_avg\": %.3f, \"TBps\": %.2f}\n", bytes / 1e9, U, NT ? "true" : "false", blocks_per_cu, sum / reps,
         bytes / (sum / reps * 1e-3) / 1e12);
  hipFree(p);
  hipFree(out);
}

template <int NACC>
static void run(int waves_per_cu, int iters) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = waves_per_cu * 64;
  mfma_loop<NACC><<<256, threads>>>(out, iters / 10, 1e-3f, 1e-3f);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0.f;
  const int reps = 5;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<256, threads>>>(out, iters, 1e-3f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double flops = 256.0 * waves_per_cu * (double)iters * 16 * 4096.0;
  const double tf_best = flops / (best * 1e-3) / 1e12, tf_avg = flops / (sum / reps * 1e-3) / 1e12;
  // one 32x32x2 fp32 MFMA occupies a SIMD's matrix pipe for 64 cycles => 256 FLOP/cycle/CU
  printf("{\"kernel\": \"mfma_f32_32x32x2 x%d acc\", \"waves_per_cu\": %d, \"ms_avg\": %.3f, \"tflops_avg\": %.1f, "
         "\"tflops_best\": %.1f, \"implied_clock_ghz\": %.3f}\n",
         NACC, waves_per_cu, sum / reps, tf_avg, tf_best, tf_avg * 1e12 / (256.0 * 256.0) / 1e9);
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'r') {  // `hw_probe r`: only the HBM read probe
    const size_t bytes = 2560000000ull;  // K1's message stream at c3 size
    run_read<4, false>(bytes, 8);
    run_read<4, true>(bytes, 8);
    run_read<8, true>(bytes, 8);
    run_read<8, true>(bytes, 4);
    run_read<4, true>(bytes, 16);
    run_read<8, false>(bytes, 8);
    return 0;
  }
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;  // ~0.1 s of MFMA per launch
  run<4>(4, iters);
  run<4>(8, iters);
  run<2>(8, iters);
  run<1>(8, iters);
  run<1>(4, iters);
  // same wall time for both roles when run together: 16 MFMA = 1024 cycles ~ 64 FMA x 4 cycles x 4
  run_coissue(20000, 0, 0);
  run_coissue(0, 80000, 0);
  run_coissue(20000, 80000, 0);
  run_coissue(20000, 20000, 0);
  run_coissue(0, 20000, 1);
  run_coissue(20000, 20000, 1);
  run_coissue(0, 20000, 2);                  // scalar ALU alone
  run_coissue(20000, 20000, 2);              // scalar ALU next to an MFMA stream
  run_coissue(20000, 20000, 0 | (1 << 4));   // MFMA mate with 4 independent accumulators
  run_coissue(20000, 20000, 0 | (2 << 4));   // VALU wave at s_setprio 3
  run_coissue(20000, 20000, 0 | (3 << 4));
  run_coissue(20000, 20000, 1 | (2 << 4));
  run_interleave<0>(20000, 4);
  run_interleave<4>(20000, 4);
  run_interleave<8>(20000, 4);
  run_interleave<12>(20000, 4);
  run_interleave<16>(20000, 4);
  run_interleave<8>(20000, 8);
  run_interleave<12>(20000, 8);
  run_interleave<16>(20000, 8);
  run_interleave_other<2, true>(20000);
  run_interleave_other<4, true>(20000);
  run_interleave_other<8, true>(20000);
  const size_t bytes = 2560000000ull;
  run_read<4, false>(bytes, 8);
  run_read<4, true>(bytes, 8);
  run_read<8, true>(bytes, 8);
  run_read<4, true>(bytes, 16);
  return 0;
}
