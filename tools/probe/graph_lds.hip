// graph_lds.hip -- does a kernel that needs > 64 KB of DYNAMIC LDS replay correctly from a HIP graph?
// (root cause hunt for the Chamfer-backward memory fault / auction time-outs under torch.cuda.graph replay:
//  both kernels ask for 131-147 KB of dynamic LDS after hipFuncSetAttribute(MaxDynamicSharedMemorySize).)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/graph_lds.hip -o /tmp/graph_lds && /tmp/graph_lds [dyn_kb] [static]
// Prints, for eager launch / stream-capture replay / explicit kernel-node replay: whether the kernel saw all of its
// LDS (it writes and reads back the last word of the segment and reports the segment size it was given).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch_dyn(unsigned *out, int words) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = (unsigned)i * 2654435761u;
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) acc += lds[i] == (unsigned)i * 2654435761u;
  atomicAdd(&out[blockIdx.x], acc);
  if (threadIdx.x == 0) out[64 + blockIdx.x] = __builtin_amdgcn_groupstaticsize();  // static part only; informative
}

__global__ void touch_static(unsigned *out) {
  __shared__ unsigned lds[36864];  // 144 KB static
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = (unsigned)i * 2654435761u;
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) acc += lds[i] == (unsigned)i * 2654435761u;
  atomicAdd(&out[blockIdx.x], acc);
}

int main(int argc, char **argv) {
  const int kb = argc > 1 ? atoi(argv[1]) : 140;
  const bool use_static = argc > 2 && !strcmp(argv[2], "static");
  const int words = kb * 256;
  unsigned *out = nullptr, host[128];
  CK(hipMalloc(reinterpret_cast<void **>(&out), sizeof host));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (!use_static)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(touch_dyn), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024));
  auto launch = [&]() {
    if (use_static) touch_static<<<8, 256, 0, s>>>(out);
    else touch_dyn<<<8, 256, (size_t)kb * 1024, s>>>(out, words);
  };
  auto report = [&](const char *what) -> int {
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(host, out, sizeof host, hipMemcpyDeviceToHost));
    printf("%-28s words ok per block:", what);
    for (int i = 0; i < 8; ++i) printf(" %u", host[i]);
    printf("  (want %d)\n", use_static ? 36864 : words);
    fflush(stdout);
    return 0;
  };
  CK(hipMemsetAsync(out, 0, sizeof host, s));
  launch();
  CK(hipGetLastError());
  if (report("eager")) return 1;
  // stream capture
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  CK(hipMemsetAsync(out, 0, sizeof host, s));
  launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 2; ++r) {
    CK(hipGraphLaunch(ge, s));
    if (report(r ? "capture replay 2" : "capture replay 1")) return 1;
  }
  // what the captured kernel node says about itself
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  hipGraphNode_t nodes[8];
  nn = nn > 8 ? 8 : nn;
  CK(hipGraphGetNodes(g, nodes, &nn));
  for (size_t i = 0; i < nn; ++i) {
    hipGraphNodeType ty;
    CK(hipGraphNodeGetType(nodes[i], &ty));
    if (ty == hipGraphNodeTypeKernel) {
      hipKernelNodeParams kp;
      CK(hipGraphKernelNodeGetParams(nodes[i], &kp));
      printf("kernel node: grid %u block %u sharedMemBytes %u\n", kp.gridDim.x, kp.blockDim.x, kp.sharedMemBytes);
    }
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  printf("done\n");
  return 0;
}
