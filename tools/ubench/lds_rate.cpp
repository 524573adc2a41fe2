// Diagnostic micro-benchmark: LDS instruction throughput per CU for the forms the scan kernels use.
//   ds_add_f32 (no return; 64 or 32 active lanes, consecutive addresses), ds_write_b32, ds_write_b128, ds_read_b128
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_rate.cpp -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4][64 * 36];
  float* my = lds[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 64 * 36; i += 64) my[i] = seed;
  float acc = seed;
  for (int it = 0; it < ITERS; ++it) {
    const int o = (it & 7) * 64;
    if constexpr (MODE == 0) { atomicAdd(my + o + lane, acc); }
    if constexpr (MODE == 1) { if (lane & 1) atomicAdd(my + o + lane, acc); }
    if constexpr (MODE == 2) { my[o + lane] = acc; }
    if constexpr (MODE == 3) { *reinterpret_cast<v4f*>(my + lane * 36 + (it & 7) * 4) = (v4f){acc, acc, acc, acc}; }
    if constexpr (MODE == 4) { v4f v = *reinterpret_cast<const v4f*>(my + lane * 36 + (it & 7) * 4); acc += v.x + v.w; }
    if constexpr (MODE == 5) { if (lane < 32) atomicAdd(my + o + lane, acc); }
    acc = acc * 1.0001f + 0.5f;
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + my[lane];
}
template <int MODE>
void run(const char* name, float* d) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s waves/SIMD %d  %.3f ms -> %.1f ns per wave-instruction per CU\n", name, wps, ms, ms * 1e6 / (4.0 * wps * ITERS));
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  run<0>("ds_add_f32 64 lanes", d); run<1>("ds_add_f32 odd lanes", d); run<5>("ds_add_f32 lanes 0-31", d);
  run<2>("ds_write_b32", d); run<3>("ds_write_b128", d); run<4>("ds_read_b128 (dependent)", d);
  return 0;
}
