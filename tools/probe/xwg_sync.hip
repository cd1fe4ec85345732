// Cost of one all-to-all hand-off between the G workgroups of a group through global memory
// (release store of a stamped slot, acquire polling of the G slots), per round.
//   hipcc --offload-arch=gfx950 -O3 -o xwg_sync xwg_sync.hip && ./xwg_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Slot { unsigned long long key; float x, y, z; unsigned stamp; unsigned pad[2]; };  // 32 B

template <int SCOPE>
__global__ void pingpong(Slot *slots, int G, int stride, int rounds, long long *cycles, int *fail) {
  // group = blocks {base, base+stride, ..., base+(G-1)*stride}
  const int grp = blockIdx.x / (G * stride) * stride + blockIdx.x % stride;
  const int me = (blockIdx.x / stride) % G;
  Slot *s = slots + (size_t)grp * G * 2;
  const long long t0 = wall_clock64();
  unsigned long long acc = 0;
  for (int r = 1; r <= rounds; ++r) {
    Slot *cur = s + (r & 1) * G;
    if (threadIdx.x == 0) {
      cur[me].key = (unsigned long long)r * 1000 + me + acc % 7;
      cur[me].x = (float)r;
      __hip_atomic_store(&cur[me].stamp, (unsigned)r, __ATOMIC_RELEASE, SCOPE);
    }
    if (threadIdx.x < G) {
      int spins = 0;
      while (__hip_atomic_load(&cur[threadIdx.x].stamp, __ATOMIC_ACQUIRE, SCOPE) != (unsigned)r) {
        if (++spins > (1 << 22)) { *fail = 1; break; }
      }
      acc += cur[threadIdx.x].key;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = wall_clock64() - t0 + (acc == 12345);
}

int main() {
  const int rounds = 4000;
  Slot *slots; long long *cyc; int *fail;
  hipMalloc(&slots, 1 << 20); hipMalloc(&cyc, 8 * 1024); hipMalloc(&fail, 4);
  for (int scope = 0; scope < 2; ++scope)
    for (int G : {2, 4, 8})
      for (int stride : {1, 8}) {  // stride 1: neighbours (different XCDs); stride 8: same XCD
        const int groups = 32, blocks = groups * G;
        if (blocks % (G * stride) != 0) continue;
        hipMemset(slots, 0, 1 << 20); hipMemset(fail, 0, 4);
        if (scope == 0) pingpong<__HIP_MEMORY_SCOPE_AGENT><<<blocks, 256>>>(slots, G, stride, rounds, cyc, fail);
        else pingpong<__HIP_MEMORY_SCOPE_SYSTEM><<<blocks, 256>>>(slots, G, stride, rounds, cyc, fail);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks); int f;
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        printf("scope %s G=%d stride=%d: %.2f us per round%s\n", scope ? "system" : "agent", G, stride,
               mx / 100.0 / rounds, f ? "  (SPIN LIMIT HIT)" : "");
      }
  return 0;
}
