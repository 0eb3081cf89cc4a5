// host for bank_gen.py kernels: bank_host <dir>   (loads k_a_x.hsaco, times 2048 blocks x 256 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
int main(int argc, char** argv) {
  std::string dir = argc > 1 ? argv[1] : ".";
  void* buf; hipMalloc(&buf, 4096);
  for (int a = 0; a < 4; ++a) for (int x = 0; x < 4; ++x) {
    char name[32]; snprintf(name, sizeof name, "k_%d_%d", a, x);
    std::string path = dir + "/" + name + ".hsaco";
    FILE* f = fopen(path.c_str(), "rb"); if (!f) { printf("missing %s\n", path.c_str()); continue; }
    std::vector<char> img; char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) img.insert(img.end(), tmp, tmp + n); fclose(f);
    hipModule_t m; hipFunction_t fn;
    if (hipModuleLoadData(&m, img.data()) != hipSuccess || hipModuleGetFunction(&fn, m, name) != hipSuccess) { printf("load failed %s\n", name); continue; }
    struct { void* p; } args{buf}; size_t sz = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipModuleLaunchKernel(fn, 2048, 1, 1, 256, 1, 1, 0, 0, nullptr, extra); hipDeviceSynchronize();
    hipEventRecord(e0); hipModuleLaunchKernel(fn, 2048, 1, 1, 256, 1, 1, 0, 0, nullptr, extra); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mads = 2048.0 * 256 * 2000 * 64;
    printf("multiplier bank %d, multiplicand base bank %d: %.3f ms  %.2f T mad/s\n", a, x, ms, mads / ms / 1e9);
  }
  return 0;
}
