// asm_loader.cpp -- loads the hand-scheduled VM kernels (gen_vm_asm.py -> *.hsaco, embedded by the build as
// vm_asm_blobs.inc) with the HIP module API and launches them with the same VmArgs block as vm_kernel<>.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <mutex>

#include "kernels.h"

namespace {
struct Blob { int wl, k; const unsigned char* data; size_t size; const char* name; };
#include "vm_asm_blobs.inc"   // defines: static const Blob kBlobs[]; static const int kNumBlobs;

struct Loaded { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; bool tried = false; hipError_t err = hipSuccess; int lds_static = -1; };
constexpr int kLdsPerCU = 160 * 1024;
constexpr int kMaxDevices = 16;
constexpr int kMaxBlobs = 64;
static_assert(kNumBlobs <= kMaxBlobs, "g_loaded is indexed by blob: raise kMaxBlobs when gen_vm_asm.SHAPES grows");
Loaded g_loaded[kMaxBlobs][kMaxDevices];   // a hipModule belongs to one device: load per (shape, device)
std::mutex g_mu;

int find_blob(int wl, int k) {
  for (int i = 0; i < kNumBlobs; ++i)
    if (kBlobs[i].wl == wl && kBlobs[i].k == k) return i;
  return -1;
}
}  // namespace

bool vm_asm_available(int wl, int k) { return find_blob(wl, k) >= 0; }

hipError_t launch_vm_asm(int wl, int k, const VmArgs& a, uint32_t blocks, hipStream_t st, int lds_share) {
  int i = find_blob(wl, k);
  if (i < 0) return hipErrorInvalidValue;
  int dev = 0;
  hipError_t de = hipGetDevice(&dev);
  if (de != hipSuccess) return de;
  if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    Loaded& L = g_loaded[i][dev];
    if (!L.tried) {
      L.tried = true;
      hipError_t e = hipModuleLoadData(&L.mod, kBlobs[i].data);
      if (e == hipSuccess) e = hipModuleGetFunction(&L.fn, L.mod, kBlobs[i].name);
      if (e != hipSuccess) { L.fn = nullptr; L.err = e; }
      int v = 0;
      if (L.fn && hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, L.fn) == hipSuccess && v >= 0 && v <= kLdsPerCU) L.lds_static = v;
    }
    if (!L.fn) return L.err != hipSuccess ? L.err : hipErrorInvalidValue;   // the first load error, every time
  }
  VmArgs args = a;
  size_t size = sizeof(VmArgs);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const Loaded& L = g_loaded[i][dev];
  unsigned dyn = 0;
  if (L.lds_static >= 0) {
    const int want = lds_share == 1 ? kLdsPerCU : lds_share == 2 ? kLdsPerCU / 2 + 2048 : 0;
    if (want > L.lds_static) dyn = (unsigned)(want - L.lds_static);
  }
  return hipModuleLaunchKernel(L.fn, blocks, 1, 1, VM_BLOCK, 1, 1, dyn, st, nullptr, extra);
}
