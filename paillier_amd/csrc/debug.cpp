// debug.cpp -- test hooks (include/paillier_hip_debug.h, NOT part of the drop-in boundary): raw VM programs on raw slot
// memory, and the planning predicates of plan.hpp as a C entry point for the CPU tests.
#include "engine.hpp"
#include "../../include/paillier_hip_debug.h"

extern "C" {

// Test hook: run a raw VM program on raw limb-major slot memory (host arrays of 28-bit limbs).
// mem_words = nslots * WT * nb uint32.  Used by tests/ to compare the assembly and hipcc kernels per opcode.
int pgpu_vm_debug_run(const pgpu_modulus* mod, const uint32_t* prog, size_t prog_words, uint32_t* mem_host,
                      size_t nslots, size_t nb, int use_asm, int* wt_out) {
  if (!mod || !prog || !mem_host) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = mod->ctx;
  const ModCtx& mc = mod->mc;
  if (wt_out) *wt_out = mc.WT;
  return guarded([&] {
    if (nb % VM_BLOCK) api_throw(PGPU_ERR_INVALID, "nb must be a multiple of 256");
    ctx->bind();
    ctx->reset_ws();
    size_t words = nslots * (size_t)mc.WT * nb;
    uint32_t* d = ctx->ws_t<uint32_t>(words);
    HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
    Prog p;
    p.w.assign(prog, prog + prog_words);
    p.asm_ok = true;
    bool saved = ctx->use_asm;
    ctx->use_asm = use_asm != 0;
    SegSpec s{&mc, &p, d, nullptr};
    try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
    ctx->use_asm = saved;
    HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_pair_debug_run(pgpu_ctx* ctx, const uint8_t* p_be, size_t p_len, int lanes, const uint32_t* prog, size_t prog_words,
                        uint32_t* mem_host, size_t nslots, size_t nb, uint32_t* consts_out, int* h_out) {
  if (!ctx || !p_be || !prog || !mem_host) return fail(PGPU_ERR_INVALID, "null argument");
  return guarded([&] {
    if (nb % VM_BLOCK) api_throw(PGPU_ERR_INVALID, "nb must be a multiple of 256");
    ctx->bind();
    ctx->reset_ws();
    const BigU pr = BigU::from_be(p_be, p_len);
    ModCtx mp, mp2;
    mp.init(ctx, pr);
    mp2.init(ctx, pr * pr);
    if (lanes == 3 || lanes == 6) {
      // three-digit kernel: slots are [3H][nb] (a0 | a1 | a2), the root is `p_be`; constants: one entry, the zero-extended
      // digits given in consts_out on entry are NOT used -- the test passes constants as slots.  lanes = 6: two lanes per digit
      if (lanes == 3 ? (mp.K != 1 || !vm_asm_available(mp.WT, 48)) : (mp.WT % 2 != 0 || !vm_asm_available(mp.WT / 2, 112)))
        api_throw(PGPU_ERR_UNSUPPORTED, "no three-digit kernel for this width");
      const int H = mp.WT;
      if (h_out) *h_out = H;
      ModCtx mp3;
      mp3.init(ctx, pr * pr * pr);
      mp3.upload();
      std::vector<uint32_t> kc = make_triple_kconsts(pr, H);
      if (consts_out) memcpy(consts_out, kc.data(), std::min(kc.size(), (size_t)3 * H) * 4);
      uint32_t* d_kc = ctx->upload_words(kc);
      size_t words = nslots * (size_t)3 * H * nb;
      uint32_t* d = ctx->ws_t<uint32_t>(words);
      HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
      Prog p;
      p.w.assign(prog, prog + prog_words);
      p.asm_ok = true;
      SegSpec s{&mp3, &p, d, nullptr};
      s.pair = d_kc; s.pair_n0inv = mp.n0inv; s.pair_h = H; s.pair_lanes = lanes; s.tconsts = d;   // constant c = slot c
      bool saved = ctx->use_asm;
      ctx->use_asm = true;
      try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
      ctx->use_asm = saved;
      HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    if (lanes != 1 && lanes != 2 && lanes != 4) api_throw(PGPU_ERR_INVALID, "lanes must be 1, 2, 3, 4 or 6");
    if (mp.K != 1 || mp2.WT != 2 * mp.WT ||
        !(lanes == 4 ? (mp.WT % 2 == 0 && vm_asm_available(mp.WT / 2, 64)) : vm_asm_available(mp.WT, lanes == 2 ? 32 : 16)))
      api_throw(PGPU_ERR_UNSUPPORTED, "no pair kernel for this width");
    mp2.upload();
    const int H = mp.WT;
    if (h_out) *h_out = H;
    std::vector<uint32_t> pc = make_pair_consts(pr, H);
    if (consts_out) memcpy(consts_out, pc.data(), pc.size() * 4);
    pc.push_back(0);
    uint32_t* d_pc = ctx->upload_words(pc);
    size_t words = nslots * (size_t)mp2.WT * nb;
    uint32_t* d = ctx->ws_t<uint32_t>(words);
    HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
    Prog p;
    p.w.assign(prog, prog + prog_words);
    p.asm_ok = true;
    SegSpec s{&mp2, &p, d, nullptr};
    s.pair = d_pc; s.pair_n0inv = mp.n0inv; s.pair_h = H; s.pair_lanes = lanes;
    bool saved = ctx->use_asm;
    ctx->use_asm = true;
    try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
    ctx->use_asm = saved;
    HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}


// The planning predicates (plan.hpp) by name: the protocol bodies call the same functions.  No context, no GPU.
int pgpu_plan_query(const char* what, const uint64_t* a, int nargs, int64_t* out, int nout) {
  if (!what || !out || (nargs > 0 && !a)) return fail(PGPU_ERR_INVALID, "null argument");
  auto need = [&](int na, int no) { return nargs >= na && nout >= no; };
  auto lt = [&](int i) { return plan::lanes_target((size_t)a[i]); };
  const std::string w(what);
  if (w == "triple_window_bits" && need(2, 1)) { out[0] = plan::triple_window_bits((size_t)a[0], (int)a[1]); return 1; }
  if (w == "crt3_ladder" && need(6, 4)) {
    const plan::Crt3Ladder l = plan::crt3_ladder((size_t)a[0], (int)a[1], (int)a[2], lt(3), a[4] != 0, a[5] != 0);
    out[0] = l.triple; out[1] = l.win; out[2] = l.nm5; out[3] = l.split;
    return 4;
  }
  if (w == "crt3_two" && need(4, 2)) {
    const plan::Crt3Two t = plan::crt3_two((size_t)a[0], (int)a[1], lt(2), a[3] != 0);
    out[0] = t.usable; out[1] = t.split;
    return 2;
  }
  if (w == "early_response_ok" && need(2, 1)) { out[0] = plan::early_response_ok((size_t)a[0], (int)a[1]); return 1; }
  if (w == "response_by_structure" && need(4, 1)) { out[0] = plan::response_by_structure((size_t)a[0], (size_t)a[1], (size_t)a[2], lt(3)); return 1; }
  if (w == "shared_chain_groups" && need(4, 1)) { out[0] = plan::shared_chain_groups((size_t)a[0], (int)a[1], lt(2), a[3] != 0); return 1; }
  if (w == "lanes_target" && need(2, 1)) { out[0] = (int64_t)plan::lanes_target((size_t)a[0], (uint32_t)a[1]); return 1; }
  if (w == "lds_share" && need(7, 1)) {
    out[0] = plan::lds_share((uint32_t)a[0], (uint32_t)a[1], a[2] != 0, a[3] != 0, a[4], a[5] != 0, a[6] != 0, nargs > 7 ? a[7] != 0 : true);
    return 1;
  }
  if (w == "generic_shape" && need(6, 2)) {
    const plan::GenericShape g = plan::generic_shape((int)a[0], (int)a[1], (size_t)a[2], (size_t)a[3], lt(4), a[5] != 0);
    out[0] = g.WL; out[1] = g.K;
    return 2;
  }
  if (w == "extract_beside" && need(3, 1)) { out[0] = plan::extract_beside((size_t)a[0], (size_t)a[1], lt(2)); return 1; }
  if (w == "pair_lanes_shared" && need(4, 1)) { out[0] = plan::pair_lanes_shared((size_t)a[0], lt(1), a[2] != 0, a[3] != 0, nargs > 4 && a[4] != 0); return 1; }
  if (w == "pair_lanes_2or4" && need(3, 1)) { out[0] = plan::pair_lanes_2or4((size_t)a[0], lt(1), a[2] != 0); return 1; }
  if (w == "pair_kernel_serves" && need(3, 1)) { out[0] = plan::pair_kernel_serves((size_t)a[0], lt(1), a[2] != 0); return 1; }
  if (w == "crt_pair_lanes" && need(4, 2)) {
    out[0] = plan::crt_pair_lanes((int)a[0], a[1] != 0, (size_t)a[2], lt(3));
    out[1] = plan::crt_pair_usable((int)out[0], 37, (size_t)a[2], lt(3));
    return 2;
  }
  if (w == "prime_lanes" && need(4, 1)) { out[0] = plan::prime_lanes((size_t)a[0], lt(1), a[2] != 0, a[3] != 0, nargs > 4 ? (int)a[4] : 1); return 1; }
  if (w == "crt_pair_lanes8" && need(4, 1)) { out[0] = plan::crt_pair_lanes8((size_t)a[0], lt(1), a[2] != 0, a[3] != 0, nargs > 4 ? (int)a[4] : 1); return 1; }
  if (w == "crt_triple_lanes6" && need(4, 1)) { out[0] = plan::crt_triple_lanes6((size_t)a[0], lt(1), a[2] != 0, a[3] != 0, nargs > 4 ? (int)a[4] : 1); return 1; }
  if (w == "triple_four_lanes_per_digit" && need(4, 1)) { out[0] = plan::triple_four_lanes_per_digit((size_t)a[0], lt(1), a[2] != 0, a[3] != 0, nargs > 4 ? (int)a[4] : 2); return 1; }
  if (w == "dual_pair_window_bits" && need(3, 1)) { out[0] = plan::dual_pair_window_bits((size_t)a[0], (int)a[1], a[2] != 0); return 1; }
  if (w == "pair_nm4_fits" && need(2, 1)) { out[0] = plan::pair_nm4_fits((size_t)a[0], (int)a[1]); return 1; }
  if (w == "shared_chain_pays" && need(2, 1)) { out[0] = plan::shared_chain_pays((size_t)a[0], lt(1)); return 1; }
  if (w == "dual_n3_two_ladders" && need(2, 1)) { out[0] = plan::dual_n3_two_ladders((size_t)a[0], lt(1)); return 1; }
  if (w == "triple_two_lanes_per_digit" && need(2, 1)) { out[0] = plan::triple_two_lanes_per_digit((size_t)a[0], lt(1)); return 1; }
  if (w == "perlane_table_slots" && need(2, 1)) { out[0] = plan::perlane_table_slots((int)a[0], a[1] != 0); return 1; }
  if (w == "gather_entries" && need(1, 1)) { out[0] = plan::gather_entries((int)a[0]); return 1; }
  return fail(PGPU_ERR_INVALID, "pgpu_plan_query: unknown decision '%s' or too few arguments / outputs", what);
}

}  // extern "C"
