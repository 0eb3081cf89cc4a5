#!/usr/bin/env python3
"""VGPR-bank experiment for the v_mad_u64_u32 stream of the row kernels: does the register index (mod 4) of the shared
multiplier / of the multiplicand array change the issue rate?  Emits one kernel per (multiplier bank, multiplicand base bank):
a loop of 64 accumulating multiplies t[j] += a * x[j], the shape of pass A.  Run with bank_host on the GPU box."""
import sys
out = sys.argv[1]
REP, NM = 2000, 64
for ab in range(4):
    for xb in range(4):
        name = f"k_{ab}_{xb}"
        a = 200 + ab                      # multiplier register
        xbase = 132 + xb                  # multiplicand array x[0..63]
        L = [f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".text", f".globl {name}", ".p2align 8", f".type {name},@function", f"{name}:"]
        L += [f"  v_mov_b32 v{a}, 3"] + [f"  v_mov_b32 v{xbase + j}, {j + 1}" for j in range(NM)] + [f"  v_mov_b64 v[{2*j}:{2*j+1}], 0" for j in range(NM)]
        L += ["  s_mov_b32 s4, 0", "  .p2align 6", "L_loop:"]
        L += [f"  v_mad_u64_u32 v[{2*j}:{2*j+1}], vcc, v{a}, v{xbase + j}, v[{2*j}:{2*j+1}]" for j in range(NM)]
        L += ["  s_add_u32 s4, s4, 1", f"  s_cmp_lt_u32 s4, {REP}", "  s_cbranch_scc1 L_loop"]
        # keep results alive: xor everything into v0 and store
        L += [f"  v_xor_b32 v0, v0, v{2*j}" for j in range(1, NM)]
        L += ["  s_load_dwordx2 s[2:3], s[0:1], 0x0", "  s_waitcnt lgkmcnt(0)", "  v_mov_b32 v1, 0", "  global_store_dword v1, v0, s[2:3]", "  s_endpgm"]
        L += [".rodata", ".p2align 6", f".amdhsa_kernel {name}", "  .amdhsa_group_segment_fixed_size 0", "  .amdhsa_private_segment_fixed_size 0",
              "  .amdhsa_kernarg_size 8", "  .amdhsa_user_sgpr_kernarg_segment_ptr 1", "  .amdhsa_next_free_vgpr 208", "  .amdhsa_next_free_sgpr 16",
              "  .amdhsa_accum_offset 208", "  .amdhsa_reserve_vcc 1", ".end_amdhsa_kernel",
              ".amdgpu_metadata", "---", "amdhsa.version: [1, 2]", "amdhsa.kernels:", f"  - .name: {name}", f"    .symbol: {name}.kd",
              "    .kernarg_segment_size: 8", "    .group_segment_fixed_size: 0", "    .private_segment_fixed_size: 0", "    .kernarg_segment_align: 8",
              "    .wavefront_size: 64", "    .sgpr_count: 16", "    .vgpr_count: 208", "    .max_flat_workgroup_size: 256", "    .args:",
              "      - .size: 8", "        .offset: 0", "        .value_kind: global_buffer", "        .address_space: global", "...", ".end_amdgpu_metadata"]
        open(f"{out}/{name}.s", "w").write("\n".join(L) + "\n")
print("ok")
