// ctx.cpp -- contexts of the C ABI (include/paillier_hip.h): device, stream, workspace, runtime switches, the profile of the
// last call; error text.  Part of libpaillier_hip.so (see engine.hpp for the map of the translation units).
#include "engine.hpp"

namespace pgi {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

[[noreturn]] void api_throw(int code, const std::string& m) { throw ApiError{code, m}; }

// overwrite key material before its storage is released (not elided: volatile stores)
void wipe(void* p, size_t n) {
  volatile unsigned char* v = (volatile unsigned char*)p;
  while (n--) *v++ = 0;
}

// Contexts that are alive: a key handle may outlive its context (garbage-collected bindings destroy in any order), so
// nothing dereferences a key's context pointer on the way out without looking here first.
static std::mutex g_live_mu;
static std::set<pgpu_ctx*> g_live_ctx;
bool ctx_alive(pgpu_ctx* c) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live_ctx.count(c) != 0;
}

}  // namespace pgi

extern "C" {

const char* pgpu_last_error(void) { return g_err.c_str(); }
const char* pgpu_version(void) { return "paillier_hip 0.1 (gfx950, radix-2^28 Montgomery VM)"; }

int pgpu_ctx_create(int device, void* stream, pgpu_ctx** out) {
  if (!out) return fail(PGPU_ERR_INVALID, "null out");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PGPU_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= n) return fail(PGPU_ERR_INVALID, "device %d out of range (have %d)", device, n);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(PGPU_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PGPU_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  pgpu_ctx* c = new pgpu_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  int rc = guarded([&] {
    c->bind();
    if (stream == PGPU_STREAM_NEW) {
      c->stream = nullptr;
      HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    }
  });
  if (rc != PGPU_OK) { delete c; return rc; }
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.insert(c); }
  *out = c;
  return PGPU_OK;
}

void pgpu_ctx_destroy(pgpu_ctx* ctx) {
  if (!ctx) return;
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.erase(ctx); }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete ctx;
}

int pgpu_ctx_set_flag(pgpu_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return fail(PGPU_ERR_INVALID, "null argument");
  if (strcmp(name, "asm") == 0) { ctx->use_asm = value != 0; return PGPU_OK; }
  if (strcmp(name, "pair") == 0) { ctx->use_pair = value != 0; return PGPU_OK; }
  if (strcmp(name, "triple") == 0) { ctx->use_triple = value != 0; return PGPU_OK; }
  if (strcmp(name, "shared_chain") == 0) { ctx->use_shared_chain = value != 0; return PGPU_OK; }
  if (strcmp(name, "lift") == 0) { ctx->use_lift = value != 0; return PGPU_OK; }
  if (strcmp(name, "side") == 0) { ctx->use_side = value != 0; return PGPU_OK; }
  if (strcmp(name, "background") == 0) { ctx->use_background = value != 0; return PGPU_OK; }
  if (strcmp(name, "exclusive") == 0) { ctx->use_exclusive = value != 0; return PGPU_OK; }
  if (strcmp(name, "exclusive_short") == 0) { ctx->use_exclusive_short = value != 0; return PGPU_OK; }
  if (strcmp(name, "base_early") == 0) { ctx->use_base_early = value != 0; return PGPU_OK; }
  if (strcmp(name, "late") == 0) { ctx->use_late = value != 0; return PGPU_OK; }
  if (strcmp(name, "lanes16") == 0) { ctx->use_lanes16 = value != 0; return PGPU_OK; }
  if (strcmp(name, "prime_lanes") == 0) { ctx->use_prime_lanes = value != 0; return PGPU_OK; }
  if (strcmp(name, "spread") == 0) { ctx->use_spread = value != 0; return PGPU_OK; }
  if (strcmp(name, "w74") == 0) { ctx->use_w74 = value != 0; return PGPU_OK; }
  if (strcmp(name, "exp_order") == 0) { ctx->use_exp_order = value != 0; return PGPU_OK; }
  if (strcmp(name, "struct") == 0) { ctx->use_struct = value != 0; return PGPU_OK; }
  if (strcmp(name, "lanes8") == 0) { ctx->use_lanes8 = value != 0; return PGPU_OK; }
  if (strcmp(name, "muls") == 0) { ctx->use_muls = value != 0; return PGPU_OK; }
  if (strcmp(name, "nm4") == 0) { ctx->use_nm4 = value != 0; return PGPU_OK; }
  if (strcmp(name, "early") == 0) { ctx->use_early = value != 0; return PGPU_OK; }
  if (strcmp(name, "handover") == 0) { ctx->use_handover = value != 0; return PGPU_OK; }
  if (strcmp(name, "fair") == 0) { g_wave_priorities.store(value != 0, std::memory_order_relaxed); return PGPU_OK; }   // process-wide (see above)
  if (strcmp(name, "lanes_wanted") == 0) { ctx->lanes_wanted = value > 0 ? (size_t)value : 0; return PGPU_OK; }
  if (strcmp(name, "cu_partition") == 0) {
    // value = (parts << 16) | part: confine this context's (own) stream to the part-th of `parts` equal slices of the
    // device's compute units.  Kernels of concurrently running contexts otherwise pile up on the same CUs (the dispatcher
    // starts every kernel's workgroups from the same place) and slow each other down instead of using the idle ones.
    if (!ctx->own_stream) return fail(PGPU_ERR_INVALID, "cu_partition needs a context created with PGPU_STREAM_NEW");
    const int parts = value >> 16, part = value & 0xffff;
    if (parts < 1 || part >= parts) return fail(PGPU_ERR_INVALID, "cu_partition: part out of range");
    return guarded([&] {
      ctx->bind();
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0);
      const int lo = (int)((long)ncu * part / parts), hi = (int)((long)ncu * (part + 1) / parts);
      for (int i = lo; i < hi; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      hipStream_t ns = nullptr;
      HIPCHK(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
      HIPCHK(hipStreamDestroy(ctx->stream));
      ctx->stream = ns;
      ctx->stream_cus = (uint32_t)(hi - lo);
    });
  }
  return fail(PGPU_ERR_INVALID, "unknown flag %s", name);
}

int pgpu_ctx_last_vm_asm(pgpu_ctx* ctx) { return ctx ? ctx->last_vm_asm : 0; }
int pgpu_ctx_last_vm_launches(pgpu_ctx* ctx) { return ctx ? ctx->last_vm_launches : 0; }

const char* pgpu_ctx_last_kernel(pgpu_ctx* ctx) {
  if (!ctx) return "";
  const pgpu_ctx::Ev* best = nullptr;
  for (size_t i = 0; i < ctx->evs_used; ++i)
    if (!best || ctx->evs[i].mads > best->mads) best = &ctx->evs[i];
  return best ? best->name : "";
}

int pgpu_ctx_last_profile(pgpu_ctx* ctx, double* vm_ms, int* vm_launches, double* vm_mads) {
  if (!ctx) return fail(PGPU_ERR_INVALID, "null ctx");
  return guarded([&] {
    ctx->bind();
    double ms = 0, mads = 0;
    for (size_t i = 0; i < ctx->evs_used; ++i) {
      HIPCHK(hipEventSynchronize(ctx->evs[i].b));
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, ctx->evs[i].a, ctx->evs[i].b));
      ms += t;
      mads += ctx->evs[i].mads;
      // PGPU_PROFILE_DUMP=1 (measurements): one line per profiled VM launch of the last call
      static const bool dump = [] { const char* e = getenv("PGPU_PROFILE_DUMP"); return e && atoi(e) != 0; }();
      if (dump) fprintf(stderr, "[pgpu] launch %2zu %-16s %9.3f ms %14.0f mads  %.3f of peak\n", i, ctx->evs[i].name, t, ctx->evs[i].mads,
                        t > 0 ? ctx->evs[i].mads / (t * 1e-3) / 39.3216e12 : 0.0);
    }
    if (vm_ms) *vm_ms = ms;
    if (vm_launches) *vm_launches = (int)ctx->evs_used;
    if (vm_mads) *vm_mads = mads;
  });
}

}  // extern "C"
