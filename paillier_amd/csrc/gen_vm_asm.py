#!/usr/bin/env python3
"""gen_vm_asm.py -- emits the hand-scheduled gfx950 assembly of the big-integer VM kernel
(`vm_asm_<WL>_<K>`), the hot kernel of the engine.

Same contract as the compiler-generated `vm_kernel<WL,K>` in kernels.hip (same VmArgs kernarg block, same
opcodes, same LDS/HBM layouts, bit-identical results); tests run both and compare.  Why assembly: the inner
loop is 2*WL v_mad_u64_u32 per row and nothing else should issue.  hipcc's version of it carries an s_nop
after every inline-asm multiply that feeds another one and waits for LDS on every modulus read (31 % of the
multiply issue rate measured); here the row is one hand-ordered stream:

  * pass A (t[j] += a_i*x[j]) runs DEPTH columns ahead of pass B (t[j-1] = t[j] + m*n[j]) so no multiply
    waits on the one before it;
  * K == 1: the modulus lives in SGPRs for the whole kernel (v_mad_u64_u32 takes an SGPR multiplicand), so a
    row touches LDS once (the next a_i, prefetched a row ahead);
  * K > 1: modulus limbs stream from LDS through rotating 4-limb buffers with counted lgkmcnt waits;
  * accumulators t[] (2*WL VGPRs) and the operand x[] (WL VGPRs) never move: pass B writes column j's
    result into column j-1's registers, which is the Montgomery shift.

Register file budget (WL = 74): 148 (t) + 74 (x) + ~30 = 252 <= 256 architectural VGPRs -> 2 waves/SIMD.

Supported opcodes: END LOAD STORE LOADC SQR MUL MULC MULV ADD (programs using SETOFF run on the
compiler-generated kernel).

Four generators share the VM contract (kernarg block, opcodes, slot layout):
  Gen    lanes of one wave hold the K slices of a number (K = 1, 2, 4)
  GenW   the two slices live in the same lane of two waves; LDS rings between them; symmetric squaring
  GenP   N = p^2, p known, 37 limbs: a residue is two base-p digits in ONE lane; a product is two Montgomery steps mod p
  GenQ   N = n^2 (or p^2), root known: the two digits in TWO neighbouring lanes, quotient digits by DPP
"""
import os
import sys

LB = 28
MASK = (1 << LB) - 1
BLOCK = 256
OPS = dict(END=0, LOAD=1, STORE=2, LOADC=3, SQR=4, MUL=5, MULC=6, MULV=7, ADD=8, SETOFF=9, MULCV=10, MULV5=11, MULV7=12, STORET=13, MULS=14,
           MULVT=15, MULVT5=16, MULCV7=17)


class Gen:
    def __init__(self, WL, K, depth=8):
        assert K in (1, 2, 4, 8)
        self.WL, self.K, self.WT = WL, K, WL * K
        self.NPB = BLOCK // K
        self.WLp = (WL + 3) // 4 * 4
        self.depth = depth
        self.n_sgpr = (K == 1 and WL <= 74)
        self.n_vreg = (K > 1 and WL <= 55)   # K > 1 with spare registers: the lane's modulus slice stays in VGPRs
        self.flush = (2 * self.WT + 1) > 255
        self.flush_every = 48
        self.sq_rows = (K == 1)  # dedicated squaring rows (symmetric products computed once)
        # multi-lane squaring (last-slice rows use the triangular skip): pays for K == 2 (-12.5 % multiplies; measured
        # +7.8 % on Decrypt-3072); for K == 4 the saving (6 %) is eaten by the per-row multiplier preparation (measured -7 %
        # on the 16k threshold batch), so those shapes keep the generic unrolled rows
        self.sq_rows_k = (K == 2 and WL <= 55)
        self.lines = []
        self.deferred = []
        self.name = f"vm_asm_{WL}_{K}"
        # ---- VGPR map: t[] pairs, x[], then scalars; pairs are even-aligned
        self.vX = 2 * WL
        e = 3 * WL
        singles = ["ai", "ain", "m", "t1"] + (["din", "drow"] if K == 1 else []) + \
                  (["k", "sh", "mk", "mult", "km2", "islast"] if (K == 2 and WL <= 55) else []) + \
                  (["upmask", "mb"] if K == 8 else [])
        for nm in singles:
            setattr(self, "v_" + nm, e)
            e += 1
        e = (e + 1) // 2 * 2
        self.v_y0 = e
        e += 2
        self.NBUF = 3
        self.nbuf = []
        self.v_N = None
        if self.n_vreg:
            self.v_N = e
            e += WL
        elif not self.n_sgpr:
            for i in range(self.NBUF):
                self.nbuf.append(e)
                e += 4
        for nm in ["goff", "aread", "awrite", "arow", "nbase", "isfirst", "notlast", "t2", "t3", "t4", "koff"]:
            setattr(self, "v_" + nm, e)
            e += 1
        self.v_addr = self.v_arow  # row pointer (inside a product) and global offset (between products) never overlap
        e = (e + 1) // 2 * 2
        self.v_p0 = e  # 64-bit temp pair
        e += 2
        self.v_p1 = e  # 64-bit temp pair; doubles as the row carry c
        self.v_c = e
        e += 2
        self.n_vgpr = e
        assert e <= 256, f"VGPR budget exceeded: {e}"
        # ---- SGPR map
        self.s_N = 20  # K == 1: modulus limbs s[20 : 20+WL)
        self.s_sbase = 94  # 64-bit scalar base for global accesses
        self.s_t0, self.s_t1 = 96, 97
        self.n_sgpr_count = 102
        # ---- LDS layout
        self.lds_n = 0
        self.lds_a = (K * self.WLp * 4 + 15) // 16 * 16 if not self.n_sgpr else 0
        self.lds_bytes = self.lds_a + (self.WT + 1) * self.NPB * 4

    # ---- helpers
    def e(self, s="", raw=False):
        self.lines.append(s if raw or not s or s.endswith(":") else "  " + s)

    def T(self, j):
        return f"v[{2 * j}:{2 * j + 1}]"

    def Tlo(self, j):
        return f"v{2 * j}"

    def Thi(self, j):
        return f"v{2 * j + 1}"

    def X(self, j):
        return f"v{self.vX + j}"

    def P(self, r):
        return f"v[{r}:{r + 1}]"

    def V(self, r):
        return f"v{r}"

    def align8(self):
        """Pin the following 8-byte instruction stream to an 8-byte boundary (the assembler pads code with s_nop).
        Hand-written v_mad_u64_u32 streams lose >10 % when they sit at 4 mod 8 (MI355X_MICROARCH.md, code placement;
        measured here: Decrypt-3072 294k -> 343k/s from placement alone)."""
        self.e(".p2align 3")

    # lane exchanges between the K slices of a number.  K <= 4: the slices sit in one quad (quad_perm).  K == 8: two quads of a
    # 16-lane DPP row -- neighbours by row_shl / row_shr (what crosses into another number is masked by isfirst / notlast, as
    # for the smaller shapes), the broadcast of slice 0 in two steps (its quad, then the quad above takes it from 4 lanes down).
    def dpp_from_next(self):
        return {2: "quad_perm:[1,1,3,3]", 4: "quad_perm:[1,2,3,3]", 8: "row_shl:1"}[self.K]

    def dpp_from_prev(self):
        return {2: "quad_perm:[0,0,2,2]", 4: "quad_perm:[0,0,1,2]", 8: "row_shr:1"}[self.K]

    def emit_bcast0(self, m):
        """m <- m of slice 0 of the same number (2 wait states since m was written are the caller's business)"""
        g, e = self, self.e
        if self.K == 8:
            e(f"v_mov_b32_dpp {m}, {m} quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf")
            e("s_nop 1")
            e(f"v_mov_b32_dpp v{g.v_mb}, {m} row_shr:4 row_mask:0xf bank_mask:0xf")
            e(f"v_bfi_b32 {m}, v{g.v_upmask}, v{g.v_mb}, {m}")          # slices 4..7 take it from 4 lanes down
        else:
            bc = {2: "[0,0,2,2]", 4: "[0,0,0,0]"}[self.K]
            e(f"v_mov_b32_dpp {m}, {m} quad_perm:{bc} row_mask:0xf bank_mask:0xf")

    def mad(self, dst, a, b, c):
        self.e(f"v_mad_u64_u32 {dst}, vcc, {a}, {b}, {c}")

    def select_segment(self):
        """Up to three program segments per launch (VmArgs: seg[3], then seg0_blocks, seg1_blocks): blocks [0, b0) run segment
        0, [b0, b0 + b1) segment 1, the rest segment 2.  Leaves s[0:1] at this block's VmSeg and s2 = block index inside it."""
        e = self.e
        e(f"s_load_dwordx2 s[4:5], s[0:1], {hex(3 * 48)}")     # seg0_blocks, seg1_blocks
        e("s_waitcnt lgkmcnt(0)")
        e("s_cmp_ge_u32 s2, s4")
        e("s_cbranch_scc0 L_seg0")
        e("s_sub_u32 s2, s2, s4")                               # block index inside segment 1 (or beyond)
        e("s_add_u32 s0, s0, 48")
        e("s_addc_u32 s1, s1, 0")
        e("s_cmp_ge_u32 s2, s5")
        e("s_cbranch_scc0 L_seg0")
        e("s_sub_u32 s2, s2, s5")                               # block index inside segment 2
        e("s_add_u32 s0, s0, 48")
        e("s_addc_u32 s1, s1, 0")
        e("L_seg0:")

    # ---------------------------------------------------------------------------------------------
    def prologue(self):
        g = self
        e = self.e
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        # s[0:1] kernarg, s2 workgroup id, v0 workitem id
        self.select_segment()
        self.timing_begin()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")   # prog, nmod, consts, mem
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")  # digits, n0inv, nb
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")  # s3 = nb*4 : byte stride between limb rows
        # lane indices
        K, WL, NPB = self.K, self.WL, self.NPB
        sh = {1: 0, 2: 1, 4: 2, 8: 3}[K]
        e(f"v_and_b32 v{g.v_t1}, {K - 1}, v0")          # k
        e(f"v_lshrrev_b32 v{g.v_t2}, {sh}, v0")          # gl
        # g = blk*NPB + gl ; goff = (k*WL*nb + g)*4
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")              # WL*nb
        e(f"v_mul_lo_u32 v{g.v_t4}, v{g.v_t1}, s{g.s_t1}")  # k*WL*nb
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_t4}, v{g.v_t3}, 2")
        # LDS addresses
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        if self.lds_a:
            e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_t4}, {WL * NPB * 4}, v{g.v_t1}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_t4}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t1}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t1}")  # byte offset of this lane's slice inside a WT-limb constant
        # masks
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_cmp_ne_u32 vcc, {K - 1}, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_notlast}, 0, -1, vcc")
        if K == 8:
            e(f"v_cmp_le_u32 vcc, 4, v{g.v_t1}")
            e("s_nop 1")
            e(f"v_cndmask_b32 v{g.v_upmask}, 0, -1, vcc")
        if self.sq_rows_k:
            e(f"v_mov_b32 v{g.v_k}, v{g.v_t1}")
            e(f"v_not_b32 v{g.v_islast}, v{g.v_notlast}")
            e(f"v_cmp_eq_u32 vcc, {K - 2}, v{g.v_t1}")
            e("s_nop 1")
            e(f"v_cndmask_b32 v{g.v_km2}, 0, -1, vcc")
        if self.n_sgpr:
            # modulus -> SGPRs s[20: 20+WL)
            off, s = 0, self.s_N
            rem = WL
            while rem > 0:
                for cnt in (16, 8, 4, 2, 1):
                    align = 4 if cnt >= 4 else cnt
                    if cnt <= rem and s % align == 0:
                        if cnt == 1:
                            e(f"s_load_dword s{s}, s[6:7], {hex(off)}")
                        else:
                            e(f"s_load_dwordx{cnt} s[{s}:{s + cnt - 1}], s[6:7], {hex(off)}")
                        off += 4 * cnt
                        s += cnt
                        rem -= cnt
                        break
                else:
                    raise RuntimeError("cannot tile the modulus into SGPR loads")
            e("s_waitcnt lgkmcnt(0)")
        else:
            # modulus -> LDS, slice k at byte offset k*WLp*4 (16-byte aligned for ds_read_b128)
            e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")                 # tid*4
            e(f"v_cmp_gt_u32 vcc, {WL}, v0")
            e("s_nop 1")
            e("s_and_saveexec_b64 s[96:97], vcc")
            for sgi in range(K):
                e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
                e("s_waitcnt vmcnt(0)")
                e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
            e("s_waitcnt lgkmcnt(0)")
            e("s_mov_b64 exec, s[96:97]")
            e("s_barrier")
            if self.n_vreg:
                for j in range(0, WL, 4):
                    cnt = min(4, WL - j)
                    if cnt == 4:
                        pass
                for j in range(WL):
                    e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
                e("s_waitcnt lgkmcnt(0)")
        # x = 0
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")

    def fair_share(self):
        """Issue arbitration between the waves of a SIMD is oldest-first: of two co-resident waves running the same long
        program the older one takes every slot it can use and the younger one only the bubbles it leaves (~7 % of the slots),
        so the older wave finishes after 55 % of the launch and the younger then runs ALONE for the other 45 % -- and a single
        wave cannot fill the SIMD (it issues at ~93 % of the rate two interleaved waves reach).  Measured with the wave
        timeline of tools/wave_timeline.py: 1 024 waves of a 2 048-wave launch end at 15.6 ms, the other 1 024 at 28.4 ms.  A
        launch whose workgroups are all resident from the start -- 65 536 ciphertexts are exactly two waves per SIMD -- loses
        ~3 % that way.  So a wave's priority FALLS with its progress: the host writes 3, 2, 1, 0 into the top two bits of the
        instruction words of the four quarters of a long program (Prog::end), and the wave that leads enters the next quarter
        with the lower priority, lets the other one catch up, and so on: they finish within a fraction of a quarter of each
        other.  (s_setprio takes an immediate: four-way branch.)"""
        if not getattr(self, "fair", True):
            return
        e = self.e
        e("s_lshr_b32 s18, s16, 30")
        for k in range(3):
            e(f"s_cmp_lg_u32 s18, {k}")
            e(f"s_cbranch_scc1 L_prio{k + 1}")
            e(f"s_setprio {k}")
            e("s_branch L_prio_done")
            e(f"L_prio{k + 1}:")
        e("s_setprio 3")
        e("L_prio_done:")

    def timing_begin(self):
        """Debug builds (PGPU_GEN_TIMING=1, K == 1 kernels, tools/wave_timeline.py): every wave notes when it started ..."""
        if not getattr(self, "timing", False):
            return
        e = self.e
        e("s_memrealtime s[100:101]")
        e("v_readfirstlane_b32 s99, v0")
        e("s_lshr_b32 s99, s99, 6")
        e("s_lshl_b32 s98, s2, 2")
        e("s_add_u32 s99, s99, s98")           # wave index inside the segment
        e("s_waitcnt lgkmcnt(0)")

    def end_of_program(self):
        """... and, at END, writes (start, end) of the 100 MHz real-time counter, HW_ID and XCC_ID to the first words of slot 0
        (32 bytes per wave, lane 0 only).  Product builds end the program here."""
        e = self.e
        if getattr(self, "timing", False):
            e("s_memrealtime s[96:97]")
            e("s_getreg_b32 s98, hwreg(HW_REG_HW_ID)")
            e("s_getreg_b32 s18, hwreg(20, 0, 32)")       # XCC_ID
            e("s_waitcnt lgkmcnt(0)")
            e("s_mov_b64 exec, 1")
            for i, sr in enumerate(("s100", "s101", "s96", "s97", "s98", "s18")):
                e(f"v_mov_b32 v{i}, {sr}")
            e("s_lshl_b32 s99, s99, 5")
            e("v_mov_b32 v6, s99")
            e("global_store_dwordx4 v6, v[0:3], s[10:11]")
            e("global_store_dwordx2 v6, v[4:5], s[10:11] offset:16")
            e("s_waitcnt vmcnt(0)")
        e("s_endpgm")

    # scalar base of a memory slot: s[sbase] = mem + arg * WT*nb*4
    def slot_base(self):
        g = self
        e = self.e
        e(f"s_mul_i32 s{g.s_t0}, s3, {self.WT}")            # WT*nb*4 (< 2^32, enforced by the host)
        e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, s{g.s_t0}")
        e(f"s_mul_i32 s{g.s_sbase}, s17, s{g.s_t0}")
        e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s10")
        e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s11")

    def const_base(self):
        g = self
        e = self.e
        e(f"s_mul_i32 s{g.s_t0}, s17, {self.WT * 4}")
        e(f"s_add_u32 s{g.s_sbase}, s8, s{g.s_t0}")
        e(f"s_addc_u32 s{g.s_sbase + 1}, s9, 0")

    def load_slot_into(self, regs):
        """regs[j] <- limb j of mem[slot] for this lane (slot base already in s[sbase])"""
        g = self
        e = self.e
        e("s_waitcnt vmcnt(0)")     # STORE does not wait for its write acknowledgements: whoever reads memory next does
        e(f"v_mov_b32 v{g.v_addr}, v{g.v_goff}")
        for j in range(self.WL):
            e(f"global_load_dword {regs[j]}, v{g.v_addr}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            if j != self.WL - 1:
                e(f"v_add_u32 v{g.v_addr}, s3, v{g.v_addr}")
        e("s_waitcnt vmcnt(0)")

    def load_const_into(self, regs):
        g = self
        for j in range(self.WL):
            self.e(f"global_load_dword {regs[j]}, v{g.v_koff}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
        self.e("s_waitcnt vmcnt(0)")

    def mask_digit_lanes(self, on):
        """hook: kernels with helper lanes (GenQ3) restrict stores / LDS staging to the lanes that own a digit"""
        pass

    def stage_to_lds(self, regs):
        """a-operand column of this lane's slice <- regs"""
        g = self
        e = self.e
        self.mask_digit_lanes(True)
        stride = self.NPB * 4
        base_reg = g.v_awrite
        cur_base_off = 0
        for j in range(self.WL):
            off = j * stride - cur_base_off
            if off > 65535:
                cur_base_off = j * stride
                e(f"v_add_u32 v{g.v_t1}, {cur_base_off}, v{g.v_awrite}")
                base_reg = g.v_t1
                off = 0
            e(f"ds_write_b32 v{base_reg}, {regs[j]} offset:{off}")
        e("s_waitcnt lgkmcnt(0)")
        self.mask_digit_lanes(False)

    # ---------------------------------------------------------------------------------------------
    def dispatcher(self):
        g = self
        e = self.e
        Xs = [self.X(j) for j in range(self.WL)]
        St = [f"v{j}" for j in range(self.WL)]  # staging registers (the accumulators are dead between products)
        e("L_next:")
        e("s_load_dwordx2 s[16:17], s[4:5], 0x0")
        e("s_add_u32 s4, s4, 8")
        e("s_addc_u32 s5, s5, 0")
        e("s_waitcnt lgkmcnt(0)")
        self.fair_share()
        e("s_and_b32 s18, s16, 0xff")
        # MULV7 / STORET (number-major tables) exist on the three-digit kernels only; the host emits them nowhere else
        nm_tables = ("MULV7", "MULVT5", "STORET") if getattr(self, "number_major_tables", False) else ()
        # MULVT / MULVT5 / STORET (number-major tables with 4- / 5-bit windows): the pair kernels GenP (37-limb primes), GenQ, GenQ4
        nm4 = ("MULVT", "MULVT5", "STORET") if getattr(self, "nm4_tables", False) else ()
        muls = ("MULS",) if getattr(self, "has_muls", False) else ()
        for nm in ("SQR", "MUL", "MULC", "MULV", "MULV5") + nm_tables + nm4 + muls + ("MULCV", "MULCV7", "LOAD", "STORE", "LOADC", "ADD"):
            e(f"s_cmp_eq_u32 s18, {OPS[nm]}")
            e(f"s_cbranch_scc1 L_{nm.lower()}")
        self.end_of_program()  # END (and anything unsupported: the host never sends those)

        e("L_load:")
        self.slot_base()
        self.load_slot_into(Xs)
        e("s_branch L_next")

        e("L_loadc:")
        self.const_base()
        self.load_const_into(Xs)
        e("s_branch L_next")

        e("L_store:")
        self.slot_base()
        self.mask_digit_lanes(True)
        e(f"v_mov_b32 v{g.v_addr}, v{g.v_goff}")
        for j in range(self.WL):
            e(f"global_store_dword v{g.v_addr}, {Xs[j]}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            if j != self.WL - 1:
                e(f"v_add_u32 v{g.v_addr}, s3, v{g.v_addr}")
        # no wait for the write acknowledgements (1 - 2 us with nothing to hide them at one wave per SIMD): store data leave
        # the registers at issue, every path that reads memory starts with s_waitcnt vmcnt(0) or waits before its dependent
        # loads (the per-number gathers), and outstanding stores complete after s_endpgm
        self.mask_digit_lanes(False)
        e("s_branch L_next")

        e("L_add:")
        self.slot_base()
        self.load_slot_into(St)
        for j in range(self.WL):
            e(f"v_add_u32 {Xs[j]}, {Xs[j]}, {St[j]}")
        e("s_branch L_next")

        e("L_mul:")
        self.slot_base()
        self.load_slot_into(St)
        self.stage_to_lds(St)
        e("s_branch L_montmul")

        if muls:
            # mem[arg] <- x * mem[arg]: the product lands in the registers of the multiplicand copy and goes straight back to the
            # slot (its base stays in s[sbase] through the rows); x is not touched
            e("L_muls:")
            self.slot_base()
            self.load_slot_into(St)
            self.stage_to_lds(St)
            e("s_branch L_montmuls")

        for lbl, per_word, wbits in (("L_mulv", 7, 4), ("L_mulv5", 5, 5)):
            # per-number table index: window `arg` of this number's own exponent -- 4 bits, 7 per 28-bit limb (MULV), or 5
            # bits, 5 per 25-bit word of the repacked exponent (MULV5); table slot = aux + digit.  Uniform control flow,
            # per-lane gather address.
            e(f"{lbl}:")
            e(f"s_mul_hi_u32 s{g.s_t1}, s17, {((1 << 32) + per_word - 1) // per_word}")   # q = arg / per_word
            e(f"s_mul_i32 s98, s{g.s_t1}, {per_word}")
            e("s_sub_u32 s98, s17, s98")                                       # r = arg % per_word
            e(f"s_mul_i32 s98, s98, {wbits}")                                  # shift = wbits * r
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s{g.s_t1}, s3")                 # digits + q * nb*4
            e(f"s_mul_i32 s{g.s_sbase}, s{g.s_t1}, s3")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s12")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s13")
            e(f"v_subrev_u32 v{g.v_t2}, {self.lds_a}, v{g.v_aread}")           # gl*4
            e(f"s_mul_i32 s{g.s_t0}, s2, {self.NPB * 4}")
            e(f"v_add_u32 v{g.v_t2}, s{g.s_t0}, v{g.v_t2}")                    # g*4
            e(f"global_load_dword v{g.v_t3}, v{g.v_t2}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            e("s_waitcnt vmcnt(0)")
            e(f"v_lshrrev_b32 v{g.v_t3}, s98, v{g.v_t3}")
            e(f"v_and_b32 v{g.v_t3}, {(1 << wbits) - 1}, v{g.v_t3}")           # digit
            e(f"s_mul_i32 s{g.s_t0}, s3, {self.WT}")                           # slot stride in bytes
            e(f"v_mul_lo_u32 v{g.v_t3}, v{g.v_t3}, s{g.s_t0}")                  # digit * stride (host guarantees < 2^32)
            e("s_bfe_u32 s17, s16, 0x160008")                                  # aux = first table slot (bits 8..29)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, s{g.s_t0}")
            e(f"s_mul_i32 s{g.s_sbase}, s17, s{g.s_t0}")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s10")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s11")
            e(f"v_add_u32 v{g.v_addr}, v{g.v_t3}, v{g.v_goff}")
            for j in range(self.WL):
                e(f"global_load_dword {St[j]}, v{g.v_addr}, s[{g.s_sbase}:{g.s_sbase + 1}]")
                if j != self.WL - 1:
                    e(f"v_add_u32 v{g.v_addr}, s3, v{g.v_addr}")
            e("s_waitcnt vmcnt(0)")
            self.stage_to_lds(St)
            e("s_branch L_montmul")

        if nm_tables:
            self.number_major_ops(St)
        if nm4:
            self.number_major4_ops(St)

        for lbl, per_word, wbits in (("L_mulcv", 7, 4), ("L_mulcv7", 4, 7)):
            # fixed-base comb: a <- consts[aux + 2^wbits * arg + digit], digit = window `arg` of this number's exponent: 4 bits, 7 per
            # 28-bit limb (MULCV), or 7 bits, 4 per limb (MULCV7: 128-entry tables, 1.75 x fewer products)
            e(f"{lbl}:")
            if per_word == 4:
                e(f"s_lshr_b32 s{g.s_t1}, s17, 2")                              # q = arg / 4
                e("s_and_b32 s98, s17, 3")
                e("s_mul_i32 s98, s98, 7")                                      # shift = 7 (arg % 4)
            else:
                e(f"s_mul_hi_u32 s{g.s_t1}, s17, {((1 << 32) + 6) // 7}")       # q = arg / 7
                e(f"s_mul_i32 s98, s{g.s_t1}, 7")
                e("s_sub_u32 s98, s17, s98")
                e("s_lshl_b32 s98, s98, 2")                                     # shift = 4 (arg % 7)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s{g.s_t1}, s3")
            e(f"s_mul_i32 s{g.s_sbase}, s{g.s_t1}, s3")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s12")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s13")
            e(f"v_subrev_u32 v{g.v_t2}, {self.lds_a}, v{g.v_aread}")
            e(f"s_mul_i32 s{g.s_t0}, s2, {self.NPB * 4}")
            e(f"v_add_u32 v{g.v_t2}, s{g.s_t0}, v{g.v_t2}")                    # g*4
            e(f"global_load_dword v{g.v_t3}, v{g.v_t2}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            e("s_waitcnt vmcnt(0)")
            e(f"v_lshrrev_b32 v{g.v_t3}, s98, v{g.v_t3}")
            e(f"v_and_b32 v{g.v_t3}, {(1 << wbits) - 1}, v{g.v_t3}")           # digit
            e(f"v_mul_u32_u24 v{g.v_t3}, {self.WT * 4}, v{g.v_t3}")            # digit * entry bytes
            e(f"v_add_u32 v{g.v_t3}, v{g.v_t3}, v{g.v_koff}")
            e("s_bfe_u32 s98, s16, 0x160008")                                  # aux (bits 8..29)
            e(f"s_lshl_b32 s17, s17, {wbits}")                                 # 2^wbits * arg
            e("s_add_u32 s17, s17, s98")
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, {self.WT * 4}")
            e(f"s_mul_i32 s{g.s_sbase}, s17, {self.WT * 4}")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s8")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s9")
            for j in range(self.WL):
                e(f"global_load_dword {St[j]}, v{g.v_t3}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
            e("s_waitcnt vmcnt(0)")
            self.stage_to_lds(St)
            e("s_branch L_montmul")

        e("L_mulc:")
        self.const_base()
        self.load_const_into(St)
        self.stage_to_lds(St)
        e("s_branch L_montmul")

        e("L_sqr:")
        if not getattr(self, "sq_self_staged", False):
            self.stage_to_lds(Xs)
        e("s_branch L_montsq" if (self.sq_rows or self.sq_rows_k or getattr(self, "sq_rows_w", False)) else "s_branch L_montmul")

    def number_major4_ops(self, St):
        """Tables that are gathered per number (the 4-bit windows of per-number exponents) stored NUMBER-major inside their slots --
        [number][WT limbs] instead of [WT limbs][number] -- by STORET and read back by MULVT: a lane's WL limbs are contiguous
        bytes, fetched with 16-byte loads that use every byte of the sectors they touch.  Limb-major, a gather reads ONE dword per
        32-byte sector (neighbouring numbers want different table entries): the multi-exponentiation of the DDLEQ prover's
        response fetched 145 GB in a 43 ms launch that way (profiles/r03_prove_traffic.txt)."""
        g, e = self, self.e
        WL = self.WL
        assert WL * 4 <= 4095 and self.vX % 2 == 0
        sfx = {4: "x4", 2: "x2", 1: ""}
        chunks, j = [], 0
        while j < WL:
            n = 4 if WL - j >= 4 else 2 if WL - j >= 2 else 1
            chunks.append((j, n))
            j += n

        def lane_offset(dst):
            """dst <- byte offset of this lane's limbs inside a number-major slot: g * WT * 4 + k * WL * 4"""
            e(f"v_subrev_u32 v{dst}, {self.lds_a}, v{g.v_aread}")            # gl*4
            e(f"s_mul_i32 s{g.s_t0}, s2, {self.NPB * 4}")
            e(f"v_add_u32 v{dst}, s{g.s_t0}, v{dst}")                        # g*4
            e(f"s_mov_b32 s{g.s_t0}, {self.WT}")
            e(f"v_mul_lo_u32 v{dst}, v{dst}, s{g.s_t0}")                     # g * WT * 4  (host guarantees < 2^32)
            if self.K > 1:
                e(f"v_add_u32 v{dst}, v{dst}, v{g.v_koff}")

        e("L_storet:")
        self.mask_digit_lanes(True)
        self.slot_base()
        lane_offset(g.v_t2)
        for j, n in chunks:
            src = self.X(j) if n == 1 else f"v[{g.vX + j}:{g.vX + j + n - 1}]"
            e(f"global_store_dword{sfx[n]} v{g.v_t2}, {src}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
        e("s_waitcnt vmcnt(0)")
        self.mask_digit_lanes(False)
        e("s_branch L_next")

        for lbl, per_word, wbits in (("L_mulvt", 7, 4), ("L_mulvt5", 5, 5)):
            # table entry = aux + the window `arg` of this number's own exponent, as MULV (4 bits, 7 per 28-bit limb) / MULV5 (5 bits,
            # 5 per 25-bit word of the repacked exponent)
            e(f"{lbl}:")
            e(f"s_mul_hi_u32 s{g.s_t1}, s17, {((1 << 32) + per_word - 1) // per_word}")   # q = arg / per_word
            e(f"s_mul_i32 s98, s{g.s_t1}, {per_word}")
            e("s_sub_u32 s98, s17, s98")
            e(f"s_mul_i32 s98, s98, {wbits}")                                  # shift = wbits * (arg % per_word)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s{g.s_t1}, s3")                 # digits + q * nb*4
            e(f"s_mul_i32 s{g.s_sbase}, s{g.s_t1}, s3")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s12")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s13")
            e(f"v_subrev_u32 v{g.v_t2}, {self.lds_a}, v{g.v_aread}")           # gl*4
            e(f"s_mul_i32 s{g.s_t0}, s2, {self.NPB * 4}")
            e(f"v_add_u32 v{g.v_t2}, s{g.s_t0}, v{g.v_t2}")                    # g*4
            e(f"global_load_dword v{g.v_t3}, v{g.v_t2}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            e(f"s_mov_b32 s{g.s_t0}, {self.WT}")
            e(f"v_mul_lo_u32 v{g.v_t2}, v{g.v_t2}, s{g.s_t0}")                 # g * WT * 4
            if self.K > 1 or getattr(self, "lanes_per_number", 1) > 1:
                e(f"v_add_u32 v{g.v_t2}, v{g.v_t2}, v{g.v_koff}")
            e("s_waitcnt vmcnt(0)")
            e(f"v_lshrrev_b32 v{g.v_t3}, s98, v{g.v_t3}")
            e(f"v_and_b32 v{g.v_t3}, {(1 << wbits) - 1}, v{g.v_t3}")           # digit
            e(f"s_mul_i32 s{g.s_t0}, s3, {self.WT}")                           # slot stride in bytes
            e(f"v_mul_lo_u32 v{g.v_t3}, v{g.v_t3}, s{g.s_t0}")                  # digit * stride (host guarantees < 2^32)
            e("s_bfe_u32 s17, s16, 0x160008")                                  # aux = first table slot (bits 8..29)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, s{g.s_t0}")
            e(f"s_mul_i32 s{g.s_sbase}, s17, s{g.s_t0}")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s10")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s11")
            e(f"v_add_u32 v{g.v_addr}, v{g.v_t3}, v{g.v_t2}")
            for j, n in chunks:
                dst = St[j] if n == 1 else f"v[{j}:{j + n - 1}]"
                e(f"global_load_dword{sfx[n]} {dst}, v{g.v_addr}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
            e("s_waitcnt vmcnt(0)")
            self.stage_to_lds(St)
            e("s_branch L_montmul")

    # ---------------------------------------------------------------------------------------------
    def gen_row(self, v_ai_cur, v_ai_next, swap_by_mov, mult_prep=None):
        """one generic row; a_i is in v_ai_next on entry (prefetched).  swap_by_mov: copy it into v_ai_cur first
        (single-body loop); otherwise the two bodies of the unrolled loop simply alternate the two registers."""
        g = self
        e = self.e
        WL, K, NPB = self.WL, self.K, self.NPB
        rstride = NPB * 4
        q = []

        def wait_for(tag):
            idx = max(i for i, t in enumerate(q) if t == tag)
            after = len(q) - 1 - idx
            e(f"s_waitcnt lgkmcnt({after})")
            del q[: idx + 1]

        e("s_waitcnt lgkmcnt(0)")
        if swap_by_mov:
            e(f"v_mov_b32 v{v_ai_cur}, v{v_ai_next}")
            ai_reg, pre_reg = v_ai_cur, v_ai_next
        else:
            ai_reg, pre_reg = v_ai_next, v_ai_cur      # use the prefetched register directly; prefetch into the other
        e(f"ds_read_b32 v{pre_reg}, v{g.v_arow}")
        q.append("ain")
        e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
        nchunks = (WL + 3) // 4
        issued = 0
        if not self.n_sgpr and not self.n_vreg:
            for cidx in range(min(g.NBUF, nchunks)):
                e(f"ds_read_b128 v[{g.nbuf[cidx % g.NBUF]}:{g.nbuf[cidx % g.NBUF] + 3}], v{g.v_nbase} offset:{16 * cidx}")
                q.append(("n", cidx))
                issued += 1

        def N(j):
            if self.n_sgpr:
                return f"s{g.s_N + j}"
            if self.n_vreg:
                return f"v{g.v_N + j}"
            return f"v{g.nbuf[(j // 4) % g.NBUF] + (j % 4)}"

        ai = f"v{ai_reg}"
        if mult_prep is not None:
            ai = mult_prep(ai)           # per-lane multiplier derived from a_i (multi-lane squaring)
        m = f"v{g.v_m}"

        def A(j):
            if j == WL - 1 and K == 1:
                self.mad(self.T(j), ai, self.X(j), "0")
            else:
                self.mad(self.T(j), ai, self.X(j), self.T(j))

        state = {"issued": issued}

        def B(j):
            if not self.n_sgpr and not self.n_vreg and j % 4 == 0:
                wait_for(("n", j // 4))
            if j == 0:
                self.mad(self.P(g.v_y0), m, N(0), self.T(0))
            elif j == 1:
                self.mad(self.T(0), m, N(1), self.T(1))
            else:
                self.mad(self.T(j - 1), m, N(j), self.T(j))
            if not self.n_sgpr and not self.n_vreg and j % 4 == 3 and state["issued"] < nchunks:
                cidx = state["issued"]
                e(f"ds_read_b128 v[{g.nbuf[cidx % g.NBUF]}:{g.nbuf[cidx % g.NBUF] + 3}], v{g.v_nbase} offset:{16 * cidx}")
                q.append(("n", cidx))
                state["issued"] += 1

        D = min(self.depth, WL)
        self.align8()
        for j in range(D):
            A(j)
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        if K > 1:
            e("s_nop 1")
            self.emit_bcast0(m)
            self.align8()
        nextA = D
        for j in range(WL):
            B(j)
            if j == 0:
                e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
                if K > 1:
                    e(f"v_and_b32 v{g.v_c}, v{g.v_c}, v{g.v_isfirst}")
                    e(f"v_and_b32 v{g.v_c + 1}, v{g.v_c + 1}, v{g.v_isfirst}")
            if nextA < WL:
                A(nextA)
                nextA += 1
            if j == 1:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
        if K > 1:
            fn = self.dpp_from_next()
            e(f"v_mov_b32_dpp {self.Tlo(WL - 1)}, v{g.v_y0} {fn} row_mask:0xf bank_mask:0xf")
            e(f"v_mov_b32_dpp {self.Thi(WL - 1)}, v{g.v_y0 + 1} {fn} row_mask:0xf bank_mask:0xf")
            e(f"v_and_b32 {self.Tlo(WL - 1)}, {self.Tlo(WL - 1)}, v{g.v_notlast}")
            e(f"v_and_b32 {self.Thi(WL - 1)}, {self.Thi(WL - 1)}, v{g.v_notlast}")

    def montmul(self):
        g = self
        e = self.e
        WL, K, NPB = self.WL, self.K, self.NPB
        rstride = NPB * 4
        e("L_montmul:")
        for j in range(WL if K > 1 else WL - 1):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
        e(f"s_mov_b32 s19, 0")
        unroll2 = (self.WT % 2 == 0)

        e(".p2align 6")
        e("L_row:")
        if unroll2:
            # two row bodies alternate the a_i registers (no copy) and halve the taken back edges
            self.gen_row(g.v_ai, g.v_ain, False)
            self.gen_row(g.v_ain, g.v_ai, False)
            e("s_add_u32 s19, s19, 2")
        else:
            self.gen_row(g.v_ai, g.v_ain, True)
            e("s_add_u32 s19, s19, 1")
        if self.flush:
            self.flush_block()
        e(f"s_cmp_lt_u32 s19, {self.WT}")
        e("s_cbranch_scc1 L_row")
        e("s_waitcnt lgkmcnt(0)")  # the dangling prefetch of row WT
        if K == 1:
            # the top column is re-created by A(WL-1) every row (src2 = 0); after the last shift it is logically 0
            e(f"v_mov_b64 {self.T(WL - 1)}, 0")
        self.normalize()
        e("s_branch L_next")
        self.lines.extend(self.deferred)
        self.deferred = []

    # ---------------------------------------------------------------------------------------------
    def montsq(self):
        """x <- x*x*R^-1 for K == 1, using the symmetry of the square:
             x^2 = sum_i x_i^2 2^(56 i) + 2 * sum_{i<j} x_i x_j 2^(28 (i+j)).
        Row i adds the doubled off-diagonal products 2 x_i x_j only for j > i (a computed jump into the pass-A
        table skips the first i+1 entries: every entry is one 8-byte v_mad_u64_u32).  The diagonal x_i^2 belongs to
        column 2i; it is added when that column sits in accumulator 0, i.e. at row 2i (even rows; x_i re-read from
        the LDS column), and for columns >= WT after the last row (static registers).  Column i is complete when
        row i starts, so the quotient digit m is computed first and its latency hides behind pass A.
        Multiplies per product: WL(WL-1)/2 + WL + WL^2 instead of 2 WL^2."""
        g = self
        e = self.e
        WL, WT = self.WL, self.WT
        assert self.K == 1 and self.n_sgpr
        rstride = self.NPB * 4
        ai2, m, din = f"v{g.v_ai}", f"v{g.v_m}", f"v{g.v_din}"
        e("L_montsq:")
        for j in range(WL):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"v_mov_b32 v{g.v_drow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"ds_read_b32 {din}, v{g.v_drow}")
        e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_drow}, {rstride}, v{g.v_drow}")
        e("s_mov_b32 s19, 0")
        N = lambda j: f"s{g.s_N + j}"

        def row_body(tag, parity):
            """one squaring row; parity: 0 even (diagonal), 1 odd, None decide at run time (odd WT shapes)"""
            e("s_waitcnt lgkmcnt(0)")
            e(f"v_add_u32 {ai2}, v{g.v_ain}, v{g.v_ain}")        # 2 * x_i
            if parity is None:
                e("s_bitcmp1_b32 s19, 0")
                e(f"s_cbranch_scc1 L_sq_odd{tag}")
            if parity in (0, None):
                self.mad(self.T(0), din, din, self.T(0))          # diagonal x_(i/2)^2 into column i
                e(f"ds_read_b32 {din}, v{g.v_drow}")
                e(f"v_add_u32 v{g.v_drow}, {rstride}, v{g.v_drow}")
            if parity is None:
                e(f"L_sq_odd{tag}:")
            e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
            e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
            e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
            e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
            # computed jump: enter the pass-A table at entry i+1 (table starts at A_1).  s[6:7] = address of L_sq_base
            # (taken once per product), s18 = 8 * s19
            e(f"s_add_u32 s98, s18, L_sqA{tag}-L_sq_base{'+8' if parity == 1 else ''}")   # odd body: row s19 + 1
            e("s_add_u32 s96, s6, s98")
            e("s_addc_u32 s97, s7, 0")
            e("s_setpc_b64 s[96:97]")
            self.align8()
            e(f"L_sqA{tag}:")
            for j in range(1, WL):
                self.mad(self.T(j), ai2, self.X(j), self.T(j))
            self.mad(self.P(g.v_y0), m, N(0), self.T(0))
            self.mad(self.T(0), m, N(1), self.T(1))
            e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
            for j in range(2, WL):
                self.mad(self.T(j - 1), m, N(j), self.T(j))
                if j == 4 or (WL <= 4 and j == WL - 1):
                    e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
            e(f"v_mov_b64 {self.T(WL - 1)}, 0")

        e("s_getpc_b64 s[6:7]")      # s6/s7 (modulus pointer) are dead after the prologue
        e("L_sq_base:")
        e("s_mov_b32 s18, 0")
        e(".p2align 6")
        e("L_rowsq:")
        if WT % 2 == 0:
            # even and odd rows as two bodies: no parity branch, half the back edges (taken branches cost fetch bubbles)
            row_body("_e", 0)
            row_body("_o", 1)
            e("s_add_u32 s19, s19, 2")
            e("s_add_u32 s18, s18, 16")
        else:
            row_body("", None)
            e("s_add_u32 s19, s19, 1")
            e("s_add_u32 s18, s18, 8")
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_rowsq")
        e("s_waitcnt lgkmcnt(0)")
        # diagonals of the columns >= WT: x_i^2 lands in accumulator 2i - WT
        for i in range((WT + 1) // 2, WT):
            self.mad(self.T(2 * i - WT), self.X(i), self.X(i), self.T(2 * i - WT))
        self.normalize()
        e("s_branch L_next")

    # ---------------------------------------------------------------------------------------------
    def montsq_k(self):
        """x <- x*x*R^-1 for K > 1 lanes per number (modulus slice in VGPRs), using the symmetry of the square at slice
        granularity.  Slice b of lane b holds x^(b).  For the rows whose multiplier a_i comes from slice a:
          lane b == a : plain products a_i * x^(a)_j          (the (a,a) block as an ordinary product)
          lane b >  a : doubled products 2 a_i * x^(b)_j      (each cross block computed once)
          lane b <  a : multiplier 0                           (its cross blocks were done in earlier rows)
        and for a == K-1 (the last slice) every other lane is idle, so those rows use the triangular form of the K == 1
        kernel: a computed jump skips the entries j <= i' and the diagonal x_i^2 -- whose column 2i is >= WT -- is added
        after the last row.  A-pass entries: (K - 1/2) WL^2 per lane instead of K WL^2."""
        g = self
        e = self.e
        WL, K, WT = self.WL, self.K, self.WT
        assert self.n_vreg
        rstride = self.NPB * 4
        m = f"v{g.v_m}"
        N = lambda j: f"v{g.v_N + j}"
        e("L_montsq:")
        for j in range(WL):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
        e("s_mov_b32 s19, 0")

        def prep(ai):
            e(f"v_lshlrev_b32 v{g.v_mult}, v{g.v_sh}, {ai}")
            e(f"v_and_b32 v{g.v_mult}, v{g.v_mult}, v{g.v_mk}")
            return f"v{g.v_mult}"

        for a in range(K - 1):
            # per-lane shift / mask for the rows of slice a
            e(f"v_cmp_lt_u32 vcc, {a}, v{g.v_k}")          # k > a
            e("s_nop 1")
            e(f"v_cndmask_b32 v{g.v_sh}, 0, 1, vcc")
            e(f"v_cmp_le_u32 vcc, {a}, v{g.v_k}")          # k >= a
            e("s_nop 1")
            e(f"v_cndmask_b32 v{g.v_mk}, 0, -1, vcc")
            e(".p2align 6")
            e(f"L_sqk_{a}:")
            self.gen_row(g.v_ai, g.v_ain, True, mult_prep=prep)
            e("s_add_u32 s19, s19, 1")
            if self.flush:
                self.flush_block(tag=f"_k{a}")
            e(f"s_cmp_lt_u32 s19, {(a + 1) * WL}")
            e(f"s_cbranch_scc1 L_sqk_{a}")
        # ---- rows of the last slice: triangular
        ai2 = f"v{g.v_mult}"
        e(".p2align 6")
        e("L_sqk_last:")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_add_u32 {ai2}, v{g.v_ain}, v{g.v_ain}")
        e(f"v_and_b32 {ai2}, {ai2}, v{g.v_islast}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {rstride}, v{g.v_arow}")
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        bc = {2: "[0,0,2,2]", 4: "[0,0,0,0]"}[K]
        e("s_nop 1")
        e(f"v_mov_b32_dpp {m}, {m} quad_perm:{bc} row_mask:0xf bank_mask:0xf")
        e("s_getpc_b64 s[96:97]")
        e("L_sqk_ret:")
        e(f"s_sub_u32 s98, s19, {(K - 1) * WL}")            # i' = row index inside the last slice
        e("s_lshl_b32 s98, s98, 3")
        e("s_add_u32 s96, s96, s98")
        e("s_addc_u32 s97, s97, 0")
        e("s_add_u32 s96, s96, L_sqkA-L_sqk_ret")
        e("s_addc_u32 s97, s97, 0")
        e("s_setpc_b64 s[96:97]")
        self.align8()
        e("L_sqkA:")
        for j in range(1, WL):
            self.mad(self.T(j), ai2, self.X(j), self.T(j))
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        e(f"v_and_b32 v{g.v_c}, v{g.v_c}, v{g.v_isfirst}")
        e(f"v_and_b32 v{g.v_c + 1}, v{g.v_c + 1}, v{g.v_isfirst}")
        for j in range(2, WL):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4 or (WL <= 4 and j == WL - 1):
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
        fn = {2: "[1,1,3,3]", 4: "[1,2,3,3]"}[K]
        e(f"v_mov_b32_dpp {self.Tlo(WL - 1)}, v{g.v_y0} quad_perm:{fn} row_mask:0xf bank_mask:0xf")
        e(f"v_mov_b32_dpp {self.Thi(WL - 1)}, v{g.v_y0 + 1} quad_perm:{fn} row_mask:0xf bank_mask:0xf")
        e(f"v_and_b32 {self.Tlo(WL - 1)}, {self.Tlo(WL - 1)}, v{g.v_notlast}")
        e(f"v_and_b32 {self.Thi(WL - 1)}, {self.Thi(WL - 1)}, v{g.v_notlast}")
        e("s_add_u32 s19, s19, 1")
        if self.flush:
            self.flush_block(tag="_klast")
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_sqk_last")
        e("s_waitcnt lgkmcnt(0)")
        # ---- diagonals of the last slice: x_i^2 (i = (K-1) WL + i') belongs to column 2i, final position 2i - WT
        for ip in range(WL):
            pos = (K - 2) * WL + 2 * ip
            lane, local = pos // WL, pos % WL
            if lane == K - 1:
                e(f"v_and_b32 v{g.v_t1}, {self.X(ip)}, v{g.v_islast}")
            else:
                assert lane == K - 2
                e(f"v_mov_b32_dpp v{g.v_t1}, {self.X(ip)} quad_perm:{fn} row_mask:0xf bank_mask:0xf")
                e(f"v_and_b32 v{g.v_t1}, v{g.v_t1}, v{g.v_km2}")
            self.mad(self.T(local), f"v{g.v_t1}", f"v{g.v_t1}", self.T(local))
        self.normalize()
        e("s_branch L_next")
        self.lines.extend(self.deferred)
        self.deferred = []

    def flush_block(self, tag=""):
        """every flush_every rows: push bits >= 2^28 of every accumulator one column up"""
        g = self
        e = self.e
        WL, K = self.WL, self.K
        # s19 already incremented: flush when s19 % flush_every == 0  (s19 = 48, 96, ...)
        e(f"s_mul_hi_u32 s{g.s_t0}, s19, {((1 << 32) + self.flush_every - 1) // self.flush_every}")  # s19 / fe
        e(f"s_mul_i32 s{g.s_t0}, s{g.s_t0}, {self.flush_every}")
        e(f"s_cmp_lg_u32 s{g.s_t0}, s19")
        e(f"s_cbranch_scc0 L_doflush{tag}")      # rare; the common path falls through (a taken branch costs a fetch bubble)
        e(f"L_noflush{tag}:")
        main = self.lines
        self.lines = self.deferred          # the flush body is emitted out of line, after the product
        e(f"L_doflush{tag}:")
        top = WL - 1
        if K > 1:
            # top_c = (t[top] >> 28) & notlast ; t[top] &= (MASK | ~notlast)
            e(f"v_lshrrev_b64 {self.P(g.v_p1)}, {LB}, {self.T(top)}")
            e(f"v_and_b32 v{g.v_p1}, v{g.v_p1}, v{g.v_notlast}")
            e(f"v_and_b32 v{g.v_p1 + 1}, v{g.v_p1 + 1}, v{g.v_notlast}")
            e(f"v_not_b32 v{g.v_t1}, v{g.v_notlast}")
            e(f"v_or_b32 v{g.v_t2}, {hex(MASK)}, v{g.v_t1}")
            e(f"v_and_b32 {self.Tlo(top)}, {self.Tlo(top)}, v{g.v_t2}")
            e(f"v_and_b32 {self.Thi(top)}, {self.Thi(top)}, v{g.v_t1}")
        for j in range(WL - 2, -1, -1):
            e(f"v_lshrrev_b64 {self.P(g.v_p0)}, {LB}, {self.T(j)}")
            e(f"v_lshl_add_u64 {self.T(j + 1)}, {self.T(j + 1)}, 0, {self.P(g.v_p0)}")
            e(f"v_and_b32 {self.Tlo(j)}, {hex(MASK)}, {self.Tlo(j)}")
            e(f"v_mov_b32 {self.Thi(j)}, 0")
        if K > 1:
            fp = self.dpp_from_prev()
            e("s_nop 1")
            e(f"v_mov_b32_dpp v{g.v_p0}, v{g.v_p1} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_mov_b32_dpp v{g.v_p0 + 1}, v{g.v_p1 + 1} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_not_b32 v{g.v_t1}, v{g.v_isfirst}")
            e(f"v_and_b32 v{g.v_p0}, v{g.v_p0}, v{g.v_t1}")
            e(f"v_and_b32 v{g.v_p0 + 1}, v{g.v_p0 + 1}, v{g.v_t1}")
            e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_p0)}")
        e(f"s_branch L_noflush{tag}")
        self.lines = main

    def normalize(self):
        """x_j = ((lo_j + mid_{j-1} + hi_{j-2}) & M) + ((lo_{j-1} + mid_{j-2} + hi_{j-3}) >> 28)"""
        g = self
        e = self.e
        WL, K = self.WL, self.K
        M = hex(MASK)
        if K == 1:
            # One lane owns the whole number: plain sequential carry, 3 instructions per limb (64-bit add, mask, 64-bit
            # shift) and fully canonical limbs.  Dependent VALU issue costs the same as independent issue on gfx950
            # (profiles/r01_valu_rates.txt, "1 dependent chain"), so the chain is not a latency problem.
            c = self.P(g.v_c)
            e(f"v_and_b32 {self.X(0)}, {M}, {self.Tlo(0)}")
            e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
            for j in range(1, WL):
                e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
                e(f"v_and_b32 {self.X(j)}, {M}, {self.Tlo(j)}")
                if j < WL - 1:
                    e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")
            return
        # K > 1: carry-save style normalisation (no chain across the lanes of a number)
        # rolling registers: mid_prev (t1), hi_prev (t2), hi_prev2 (t3), s_prev (t4)
        mid_p, hi_p, hi_p2, s_p = f"v{g.v_t1}", f"v{g.v_t2}", f"v{g.v_t3}", f"v{g.v_t4}"
        if K > 1:
            fp = self.dpp_from_prev()
            top, top2 = WL - 1, WL - 2
            # incoming mid_{-1}, hi_{-1}, hi_{-2} from the previous lane's top columns
            e(f"v_alignbit_b32 v{g.v_p0}, {self.Thi(top)}, {self.Tlo(top)}, {LB}")
            e(f"v_and_b32 v{g.v_p0}, {M}, v{g.v_p0}")                       # mid_top
            e(f"v_lshrrev_b32 v{g.v_p0 + 1}, {2 * LB - 32}, {self.Thi(top)}")     # hi_top
            e(f"v_lshrrev_b32 v{g.v_p1}, {2 * LB - 32}, {self.Thi(top2)}")        # hi_top2
            e(f"v_not_b32 v{g.v_p1 + 1}, v{g.v_isfirst}")
            e("s_nop 1")
            e(f"v_mov_b32_dpp {mid_p}, v{g.v_p0} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_mov_b32_dpp {hi_p}, v{g.v_p0 + 1} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_mov_b32_dpp {hi_p2}, v{g.v_p1} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_and_b32 {mid_p}, {mid_p}, v{g.v_p1 + 1}")
            e(f"v_and_b32 {hi_p}, {hi_p}, v{g.v_p1 + 1}")
            e(f"v_and_b32 {hi_p2}, {hi_p2}, v{g.v_p1 + 1}")
        else:
            e(f"v_mov_b32 {mid_p}, 0")
            e(f"v_mov_b32 {hi_p}, 0")
            e(f"v_mov_b32 {hi_p2}, 0")
        # pass 1: s_j into X(j) (x is dead after the rows)
        for j in range(WL):
            lo = f"v{g.v_p0}"
            e(f"v_and_b32 {lo}, {M}, {self.Tlo(j)}")
            e(f"v_add3_u32 {self.X(j)}, {lo}, {mid_p}, {hi_p2}")
            # roll: hi_p2 <- hi_p ; hi_p <- hi_j ; mid_p <- mid_j
            e(f"v_mov_b32 {hi_p2}, {hi_p}")
            e(f"v_lshrrev_b32 {hi_p}, {2 * LB - 32}, {self.Thi(j)}")
            e(f"v_alignbit_b32 {mid_p}, {self.Thi(j)}, {self.Tlo(j)}, {LB}")
            e(f"v_and_b32 {mid_p}, {M}, {mid_p}")
        # pass 2 (descending so s_{j-1} is still unmodified): x_j = (s_j & M) + (s_{j-1} >> 28)
        if K > 1:
            fp = self.dpp_from_prev()
            e(f"v_lshrrev_b32 v{g.v_p0}, {LB}, {self.X(WL - 1)}")
            e(f"v_not_b32 v{g.v_p1 + 1}, v{g.v_isfirst}")
            e("s_nop 1")
            e(f"v_mov_b32_dpp {s_p}, v{g.v_p0} {fp} row_mask:0xf bank_mask:0xf")
            e(f"v_and_b32 {s_p}, {s_p}, v{g.v_p1 + 1}")
        for j in range(WL - 1, -1, -1):
            if j >= 1:
                e(f"v_lshrrev_b32 v{g.v_p0}, {LB}, {self.X(j - 1)}")
                e(f"v_and_b32 {self.X(j)}, {M}, {self.X(j)}")
                e(f"v_add_u32 {self.X(j)}, {self.X(j)}, v{g.v_p0}")
            else:
                e(f"v_and_b32 {self.X(0)}, {M}, {self.X(0)}")
                if K > 1:
                    e(f"v_add_u32 {self.X(0)}, {self.X(0)}, {s_p}")

    def epilogue(self):
        e = self.e
        nm = self.name
        e(".rodata")
        e(".p2align 6")
        e(f".amdhsa_kernel {nm}")
        e(f"  .amdhsa_group_segment_fixed_size {self.lds_bytes}")
        e("  .amdhsa_private_segment_fixed_size 0")
        e("  .amdhsa_kernarg_size 152")
        e("  .amdhsa_user_sgpr_kernarg_segment_ptr 1")
        e("  .amdhsa_system_sgpr_workgroup_id_x 1")
        e("  .amdhsa_system_vgpr_workitem_id 0")
        e(f"  .amdhsa_next_free_vgpr {self.n_vgpr}")
        e(f"  .amdhsa_next_free_sgpr {self.n_sgpr_count}")
        e(f"  .amdhsa_accum_offset {(self.n_vgpr + 3) // 4 * 4}")
        e("  .amdhsa_reserve_vcc 1")
        e(".end_amdhsa_kernel")
        e(".amdgpu_metadata", raw=True)
        e("---", raw=True)
        e("amdhsa.version: [1, 2]", raw=True)
        e("amdhsa.kernels:", raw=True)
        e(f"  - .name: {nm}", raw=True)
        e(f"    .symbol: {nm}.kd", raw=True)
        e("    .kernarg_segment_size: 152", raw=True)
        e(f"    .group_segment_fixed_size: {self.lds_bytes}", raw=True)
        e("    .private_segment_fixed_size: 0", raw=True)
        e("    .kernarg_segment_align: 8", raw=True)
        e("    .wavefront_size: 64", raw=True)
        e(f"    .sgpr_count: {self.n_sgpr_count}", raw=True)
        e(f"    .vgpr_count: {self.n_vgpr}", raw=True)
        e("    .max_flat_workgroup_size: 256", raw=True)
        e("    .args:", raw=True)
        e("      - .size: 152", raw=True)
        e("        .offset: 0", raw=True)
        e("        .value_kind: by_value", raw=True)
        e("...", raw=True)
        e(".end_amdgpu_metadata", raw=True)

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        if self.sq_rows:
            self.montsq()
        if self.sq_rows_k:
            self.montsq_k()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenW(Gen):
    """Wave-sliced shape: the S = `K` slices of WL limbs of one number live in the SAME lane of S different waves of
    the workgroup (wave w: slice w % S of number group w / S), not in neighbouring lanes of one wave.  Every wave then
    runs the single-lane row (modulus slice in SGPRs, triangular squaring possible) and the slices talk through LDS:

      * the quotient digit m_i is computed by the bottom wave (slice 0) and published in a 2-slot ring;
      * the shifted-out column of an upper wave (its low 28 bits; the rest is carried inside the wave) is published in a
        2-slot ring and added by the wave below two rows later -- the upper wave runs about one row behind, nobody waits
        in steady state;
      * after the last row the final carry travels upwards once.

    Every word carries a 4-bit tag (valid | op parity | row/2 mod 4) next to its 28-bit payload, so a reader sees at
    once whether the word it read is the one it needs; a mismatch takes an out-of-line bounded spin.  Producer/consumer
    order (m_r is written after lo_{r-2} was verified; lo_i is written after m_i was verified) makes 2 slots enough.
    One s_barrier per product (after the a-operand column is staged) orders the column writes.
    """

    def __init__(self, WL, S, depth=8):
        self.fair = False          # the two waves of a number wait for each other's ring words: no priority games here
        assert S == 2, "two wave slices for now"
        Gen.__init__(self, WL, 1, depth)
        self.S = S
        self.K = 1                      # register map and row code of the single-lane shape
        self.Kabi = S                   # what the host calls K (numbers per block = 256 / S)
        self.WT = WL * S
        self.NPB = BLOCK // S
        self.name = f"vm_asm_{WL}_{S}"
        self.flush = False
        assert 3 * WL + 3 <= 255, "a column lives WL rows and takes <= 3 product units per row"
        self.sq_rows = False
        self.sq_rows_k = False
        self.sq_rows_w = True           # symmetric squaring split over the two waves (montsq below)
        # LDS: rings first, then the a-operand column [WT + 1][NPB]
        row = self.NPB * 4
        self.lds_m = 0                  # m ring: 2 slots
        self.lds_lo = 2 * row           # lo ring: 2 slots
        self.lds_cy = 4 * row           # final carry: 2 words
        self.lds_a = 6 * row
        self.lds_bytes = self.lds_a + (self.WT + 1) * row
        # extra VGPRs (the K == 1 map leaves room: nbase / isfirst / notlast are unused)
        self.v_ring = self.v_nbase      # gl * 4
        self.v_raw = self.v_isfirst     # word read from a ring
        self.v_lo = self.v_p0           # pair (payload, 0): v_p0 + 1 is kept 0 inside a product
        # SGPRs
        self.s_slice, self.s_dead, self.s_lim = 0, 1, 15
        self.s_exp, self.s_exp2, self.s_par = 99, 100, 101
        self.s_cnt = 16
        self.SPIN_LIMIT = 200000
        self.uid = 0

    # ---- tag helpers ------------------------------------------------------------------------------------------
    def set_exp(self, both=True):
        """s_exp = tag word of rows (s19, s19+1); s_exp2 = tag word of rows (s19+2, s19+3) (top wave only)"""
        g, e = self, self.e
        e(f"s_lshl_b32 s{g.s_exp}, s19, 27")               # (s19 >> 1) << 28
        e(f"s_and_b32 s{g.s_exp}, s{g.s_exp}, 0x30000000")
        if both:
            e(f"s_add_u32 s{g.s_exp2}, s{g.s_exp}, 0x10000000")
            e(f"s_and_b32 s{g.s_exp2}, s{g.s_exp2}, 0x30000000")
            e(f"s_or_b32 s{g.s_exp2}, s{g.s_exp2}, s{g.s_par}")
        e(f"s_or_b32 s{g.s_exp}, s{g.s_exp}, s{g.s_par}")

    def check(self, dst, off, exp):
        """dst <- payload of v_raw if its tag equals `exp`; else spin (bounded) re-reading LDS word v_ring + off"""
        g, e = self, self.e
        self.uid += 1
        u = self.uid
        e(f"v_xor_b32 {dst}, s{exp}, v{g.v_raw}")
        e(f"v_cmp_le_u32 vcc, s{g.s_lim}, {dst}")
        e(f"s_cbranch_vccnz L_slow{u}")
        e(f"L_ok{u}:")
        main = self.lines
        self.lines = self.deferred
        e(f"L_slow{u}:")
        e(f"s_cmp_lg_u32 s{g.s_dead}, 0")
        e(f"s_cbranch_scc1 L_ok{u}")
        e(f"s_mov_b32 s{g.s_cnt}, 0")
        e(f"L_spin{u}:")
        e("s_sleep 1")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{off}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_xor_b32 {dst}, s{exp}, v{g.v_raw}")
        e(f"v_cmp_le_u32 vcc, s{g.s_lim}, {dst}")
        e(f"s_cbranch_vccz L_ok{u}")
        e(f"s_add_u32 s{g.s_cnt}, s{g.s_cnt}, 1")
        e(f"s_cmp_lt_u32 s{g.s_cnt}, {g.SPIN_LIMIT}")
        e(f"s_cbranch_scc1 L_spin{u}")
        e(f"s_mov_b32 s{g.s_dead}, 1")                   # partner never answered: stop waiting for the rest of the kernel
        e(f"s_branch L_ok{u}")
        self.lines = main

    # ---- prologue ---------------------------------------------------------------------------------------------
    def prologue(self):
        g, e = self, self.e
        WL, NPB, S = self.WL, self.NPB, self.S
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")                                   # nb*4
        # wave -> (slice, group); s0/s1 are free from here on
        e(f"v_lshrrev_b32 v{g.v_t1}, 6, v0")
        e("s_nop 1")
        e(f"v_readfirstlane_b32 s{g.s_t0}, v{g.v_t1}")               # wave id
        e(f"s_and_b32 s{g.s_slice}, s{g.s_t0}, {S - 1}")
        e(f"s_lshr_b32 s{g.s_t0}, s{g.s_t0}, {S.bit_length() - 1}")   # group
        e(f"s_lshl_b32 s{g.s_t0}, s{g.s_t0}, 6")
        e(f"v_and_b32 v{g.v_t2}, 63, v0")
        e(f"v_add_u32 v{g.v_t2}, s{g.s_t0}, v{g.v_t2}")              # gl = group*64 + lane
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")              # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"s_mul_i32 s{g.s_t1}, s{g.s_t1}, s{g.s_slice}")           # slice*WL*nb
        e(f"v_add_u32 v{g.v_t4}, s{g.s_t1}, v{g.v_t3}")
        e(f"v_lshlrev_b32 v{g.v_goff}, 2, v{g.v_t4}")
        e(f"v_lshlrev_b32 v{g.v_ring}, 2, v{g.v_t2}")                # gl*4
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_ring}")
        e(f"s_mul_i32 s{g.s_t0}, s{g.s_slice}, {WL * NPB * 4}")
        e(f"v_add_u32 v{g.v_awrite}, s{g.s_t0}, v{g.v_aread}")
        e(f"s_mul_i32 s{g.s_t0}, s{g.s_slice}, {WL * 4}")
        e(f"v_mov_b32 v{g.v_koff}, s{g.s_t0}")
        # this wave's modulus slice -> SGPRs
        e(f"s_add_u32 s6, s6, s{g.s_t0}")
        e("s_addc_u32 s7, s7, 0")
        off, s, rem = 0, self.s_N, WL
        while rem > 0:
            for cnt in (16, 8, 4, 2, 1):
                align = 4 if cnt >= 4 else cnt
                if cnt <= rem and s % align == 0:
                    if cnt == 1:
                        e(f"s_load_dword s{s}, s[6:7], {hex(off)}")
                    else:
                        e(f"s_load_dwordx{cnt} s[{s}:{s + cnt - 1}], s[6:7], {hex(off)}")
                    off += 4 * cnt
                    s += cnt
                    rem -= cnt
                    break
            else:
                raise RuntimeError("cannot tile the modulus into SGPR loads")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mov_b32 s{g.s_lim}, 0x10000000")
        e(f"s_mov_b32 s{g.s_dead}, 0")
        e(f"s_mov_b32 s{g.s_par}, 0x80000000")                       # valid | parity 0
        # rings: m and carry words invalid (tag 0); lo ring pre-loaded with lo_{-2} = lo_{-1} = 0 for the first product
        e(f"v_mov_b32 v{g.v_t1}, 0")
        e(f"v_mov_b32 v{g.v_t3}, s{g.s_par}")
        row = NPB * 4
        for k in range(2):
            e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_m + k * row}")
            e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_cy + k * row}")
            e(f"ds_write_b32 v{g.v_ring}, v{g.v_t3} offset:{self.lds_lo + k * row}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")

    # ---- one row --------------------------------------------------------------------------------------------
    def row_bottom(self, ai_reg, pre_reg, b, full=True, mult=None, a_from=0):
        """slice 0.  b: row parity (ring slot).  The a_i for this row is in ai_reg; the next one is prefetched into
        pre_reg.  v_raw holds the ring word lo_{r-2} (read issued a row earlier)."""
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        N = lambda j: f"s{g.s_N + j}"
        m = f"v{g.v_m}"
        e("s_waitcnt lgkmcnt(0)")
        self.check(f"v{g.v_lo}", self.lds_lo + b * row, g.s_exp)
        e(f"v_lshl_add_u64 {self.T(WL - 2)}, {self.T(WL - 2)}, 0, {self.P(g.v_lo)}")
        e(f"ds_read_b32 v{pre_reg}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_lo + (1 - b) * row}")    # lo_{r-1}, needed next row
        ai = f"v{ai_reg}"
        if mult is not None:
            ai = mult(ai)
        self.align8()
        if full:
            self.mad(self.T(0), ai, self.X(0), self.T(0))
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        e(f"v_or_b32 v{g.v_t1}, s{g.s_exp}, {m}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_m + b * row}")
        if full:
            self.align8()
            for j in range(1, WL):
                if j == WL - 1:
                    self.mad(self.T(j), ai, self.X(j), "0")
                else:
                    self.mad(self.T(j), ai, self.X(j), self.T(j))
        self.align8()
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        for j in range(2, WL):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
        if not full:
            e(f"v_mov_b64 {self.T(WL - 1)}, 0")

    def row_top(self, ai_reg, pre_reg, b, full=True):
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        N = lambda j: f"s{g.s_N + j}"
        m = f"v{g.v_m}"
        e("s_waitcnt lgkmcnt(0)")
        e(f"ds_read_b32 v{pre_reg}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_m + b * row}")           # m_i
        ai = f"v{ai_reg}"
        self.align8()
        for j in range(WL):
            if j == WL - 1:
                self.mad(self.T(j), ai, self.X(j), "0")
            else:
                self.mad(self.T(j), ai, self.X(j), self.T(j))
        e("s_waitcnt lgkmcnt(0)")
        self.check(m, self.lds_m + b * row, g.s_exp)
        self.align8()
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        e(f"v_and_b32 v{g.v_t1}, {hex(MASK)}, v{g.v_y0}")
        e(f"v_or_b32 v{g.v_t1}, s{g.s_exp2}, v{g.v_t1}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_lo + b * row}")          # lo_i
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        self.align8()
        for j in range(2, WL):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")

    # ---- product ----------------------------------------------------------------------------------------------
    def product_entry(self):
        """common head of a product: the a column is staged; order it, clear the accumulators, flip the parity"""
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        e("s_barrier")
        for j in range(WL - 1):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_lo + 1}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e("s_mov_b32 s19, 0")

    def finish_bottom(self):
        """after the last row: the two outstanding lo words, sequential carry, final carry up"""
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        M = hex(MASK)
        self.set_exp()
        e("s_waitcnt lgkmcnt(0)")                         # v_raw = lo_{WT-2} (read issued in the last row), dangling a prefetch
        self.check(f"v{g.v_lo}", self.lds_lo + 0 * row, g.s_exp)
        e(f"v_lshl_add_u64 {self.T(WL - 2)}, {self.T(WL - 2)}, 0, {self.P(g.v_lo)}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_lo + 1 * row}")
        e("s_waitcnt lgkmcnt(0)")
        self.check(f"v{g.v_lo}", self.lds_lo + 1 * row, g.s_exp)
        e(f"v_mov_b32 {self.Tlo(WL - 1)}, v{g.v_lo}")
        e(f"v_mov_b32 {self.Thi(WL - 1)}, 0")
        c = self.P(g.v_c)
        e(f"v_and_b32 {self.X(0)}, {M}, {self.Tlo(0)}")
        e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
        for j in range(1, WL):
            e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            e(f"v_and_b32 {self.X(j)}, {M}, {self.Tlo(j)}")
            e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")
        # carry (< 2^37) -> two tagged words
        e(f"v_and_b32 v{g.v_t1}, {M}, v{g.v_c}")
        e(f"v_or_b32 v{g.v_t1}, s{g.s_par}, v{g.v_t1}")
        e(f"v_lshrrev_b64 {c}, {LB}, {c}")
        e(f"v_or_b32 v{g.v_t2}, s{g.s_par}, v{g.v_c}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_cy}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t2} offset:{self.lds_cy + row}")
        e(f"s_xor_b32 s{g.s_par}, s{g.s_par}, 0x40000000")
        e("s_branch L_next")

    def finish_top(self, top_zeroed=False):
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        M = hex(MASK)
        e("s_waitcnt lgkmcnt(0)")
        if not top_zeroed:
            e(f"v_mov_b64 {self.T(WL - 1)}, 0")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_cy}")
        e("s_waitcnt lgkmcnt(0)")
        self.check(f"v{g.v_lo}", self.lds_cy, g.s_par)
        e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_lo)}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_cy + row}")
        e("s_waitcnt lgkmcnt(0)")
        self.check(f"v{g.v_lo}", self.lds_cy + row, g.s_par)
        e(f"v_lshl_add_u64 {self.T(1)}, {self.T(1)}, 0, {self.P(g.v_lo)}")
        # the bottom wave is past its last lo word: pre-load lo_{-2} = lo_{-1} = 0 of the next product
        e(f"s_xor_b32 s{g.s_par}, s{g.s_par}, 0x40000000")
        e(f"v_mov_b32 v{g.v_t1}, s{g.s_par}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_lo}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_lo + row}")
        c = self.P(g.v_c)
        e(f"v_and_b32 {self.X(0)}, {M}, {self.Tlo(0)}")
        e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
        for j in range(1, WL):
            e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            e(f"v_and_b32 {self.X(j)}, {M}, {self.Tlo(j)}")
            if j < WL - 1:
                e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_branch L_next")

    def montmul(self):
        g, e = self, self.e
        WT = self.WT
        row = self.NPB * 4
        assert WT % 2 == 0
        e("L_montmul:")
        self.product_entry()
        e(f"s_cmp_eq_u32 s{g.s_slice}, 0")
        e("s_cbranch_scc0 L_mm_top")
        # ---------------- bottom wave
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_lo}")        # lo_{-2}
        e(".p2align 6")
        e("L_rowb:")
        self.set_exp(both=False)
        self.row_bottom(g.v_ain, g.v_ai, 0)
        self.row_bottom(g.v_ai, g.v_ain, 1)
        e("s_add_u32 s19, s19, 2")
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_rowb")
        self.finish_bottom()
        # ---------------- top wave
        e(".p2align 6")
        e("L_mm_top:")
        e("L_rowt:")
        self.set_exp()
        self.row_top(g.v_ain, g.v_ai, 0)
        self.row_top(g.v_ai, g.v_ain, 1)
        e("s_add_u32 s19, s19, 2")
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_rowt")
        self.finish_top()
        self.lines.extend(self.deferred)
        self.deferred = []

    # ---- squaring ---------------------------------------------------------------------------------------------
    def jump_base(self, label):
        """once per product and wave role: s[6:7] = address of `label` (s6/s7 are free after the prologue); the running
        table offset 8*idx(row s19) lives in s18"""
        e = self.e
        e("s_getpc_b64 s[6:7]")
        e(f"{label}:")

    def jump_into(self, p, table, base, reset):
        """computed jump of row s19 + p into `table` (8-byte entries) at entry idx(row): idx grows by one per row and is
        set to `reset` at row WL.  s18 = 8*idx(s19) on entry of body 0; body 1 leaves 8*idx(s19 + 1) in s17."""
        e = self.e
        WL = self.WL
        if p == 0:
            src = "s18"
        else:
            e("s_add_u32 s17, s18, 8")
            if WL % 2 == 1:
                e(f"s_cmp_eq_u32 s19, {WL - 1}")
                e(f"s_cselect_b32 s17, {8 * reset}, s17")
            src = "s17"
        e(f"s_add_u32 s98, {src}, {table}-{base}")
        e("s_add_u32 s96, s6, s98")
        e("s_addc_u32 s97, s7, 0")
        e("s_setpc_b64 s[96:97]")

    def jump_next(self, reset):
        """after `s_add_u32 s19, s19, 2`: s18 = 8*idx of the new row s19"""
        e = self.e
        e("s_add_u32 s18, s17, 8")
        if self.WL % 2 == 0:
            e(f"s_cmp_eq_u32 s19, {self.WL}")
            e(f"s_cselect_b32 s18, {8 * reset}, s18")

    def b_stream(self, publish_lo=None):
        """pass B with the shift; publish_lo: (slot offset) for the top wave"""
        g, e = self, self.e
        WL = self.WL
        N = lambda j: f"s{g.s_N + j}"
        m = f"v{g.v_m}"
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        if publish_lo is not None:
            e(f"v_and_b32 v{g.v_t1}, {hex(MASK)}, v{g.v_y0}")
            e(f"v_or_b32 v{g.v_t1}, s{g.s_exp2}, v{g.v_t1}")
            e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{publish_lo}")
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        if publish_lo is not None:
            self.align8()
        for j in range(2, WL):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
        e(f"v_mov_b64 {self.T(WL - 1)}, 0")

    def sq_row_bottom(self, p):
        """symmetric squaring row of slice 0 (row r = s19 + p).  a_r (any limb of x) times the lo slice, doubled, for the
        columns c > r mod WL; the diagonal x_(r/2)^2 on even rows; see montsq()."""
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        m, ai2, din = f"v{g.v_m}", f"v{g.v_ai}", f"v{g.v_din}"
        e("s_waitcnt lgkmcnt(0)")
        self.check(f"v{g.v_lo}", self.lds_lo + p * row, g.s_exp)
        e(f"v_lshl_add_u64 {self.T(WL - 2)}, {self.T(WL - 2)}, 0, {self.P(g.v_lo)}")
        e(f"v_add_u32 {ai2}, v{g.v_ain}, v{g.v_ain}")
        if p == 0:
            self.mad(self.T(0), din, din, self.T(0))
            e(f"ds_read_b32 {din}, v{g.v_drow}")
            e(f"v_add_u32 v{g.v_drow}, {row}, v{g.v_drow}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_lo + (1 - p) * row}")
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        e(f"v_or_b32 v{g.v_t1}, s{g.s_exp}, {m}")
        e(f"ds_write_b32 v{g.v_ring}, v{g.v_t1} offset:{self.lds_m + p * row}")
        tag = f"_sb{p}"
        self.jump_into(p, f"L_sqA{tag}", "L_sqbase_b", 0)
        self.align8()
        e(f"L_sqA{tag}:")
        for j in range(1, WL):
            self.mad(self.T(j), ai2, self.X(j), self.T(j))
        self.b_stream()

    def sq_row_top(self, p):
        """symmetric squaring row of slice 1 (row r = s19 + p).  Rows r < WL (a_r = lo limb r): doubled products with
        hi[c], c >= r.  Rows r >= WL (a_r = hi limb i' = r - WL): doubled products with hi[c], c > i', and when r - WL is
        even the diagonal hi[(r - WL)/2]^2 into column 0 (the diagonal register holds 0 until row WL)."""
        g, e = self, self.e
        WL = self.WL
        row = self.NPB * 4
        m, ai2, din = f"v{g.v_m}", f"v{g.v_ai}", f"v{g.v_din}"
        tag = f"_st{p}"
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_add_u32 {ai2}, v{g.v_ain}, v{g.v_ain}")
        if p == (WL & 1):
            self.mad(self.T(0), din, din, self.T(0))
            # next diagonal operand (for row r + 2) once r + 2 >= WL
            e(f"s_cmp_lt_u32 s19, {WL - 2 - p}")
            e(f"s_cbranch_scc1 L_nodin{tag}")
            e(f"ds_read_b32 {din}, v{g.v_drow}")
            e(f"v_add_u32 v{g.v_drow}, {row}, v{g.v_drow}")
            e(f"L_nodin{tag}:")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_m + p * row}")

        self.jump_into(p, f"L_sqA{tag}", "L_sqbase_t", 1)
        self.align8()
        e(f"L_sqA{tag}:")
        for j in range(WL):
            self.mad(self.T(j), ai2, self.X(j), self.T(j))
        e("s_waitcnt lgkmcnt(0)")
        self.check(m, self.lds_m + p * row, g.s_exp)
        self.align8()
        self.b_stream(publish_lo=self.lds_lo + p * row)

    def montsq(self):
        """x <- x*x*R^-1 with x = lo + hi B^WL split over the two waves, every product x_i x_j (i < j) computed once:
             both in lo          bottom wave, row i          (columns c = j > i)
             both in hi          top wave,    row WL + i'    (columns c = j' > i')
             i in lo, j in hi    top wave at row i when j' >= i;  bottom wave at row WL + j' when j' < i
           so every row of either wave is a triangular row (a computed jump skips the leading table entries) and the two
           waves carry the same number of multiplies in every row -- they stay in lock step.  Diagonals x_k^2 join column
           2k when it is accumulator 0 of the wave that holds it (bottom: rows 2k; top: rows WL + 2k'), the rest (top
           wave, k' >= WL/2) after the last row from the static registers."""
        g, e = self, self.e
        WL, WT = self.WL, self.WT
        row = self.NPB * 4
        e("L_montsq:")
        e("s_barrier")
        for j in range(WL):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_lo + 1}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        e("s_mov_b32 s19, 0")
        e(f"s_cmp_eq_u32 s{g.s_slice}, 0")
        e("s_cbranch_scc0 L_sq_top")
        # ---------------- bottom wave
        e(f"v_mov_b32 v{g.v_drow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_din}, v{g.v_drow}")
        e(f"v_add_u32 v{g.v_drow}, {row}, v{g.v_drow}")
        e(f"ds_read_b32 v{g.v_raw}, v{g.v_ring} offset:{self.lds_lo}")
        self.jump_base("L_sqbase_b")
        e("s_mov_b32 s18, 0")
        e(".p2align 6")
        e("L_sqb:")
        self.set_exp(both=False)
        self.sq_row_bottom(0)
        self.sq_row_bottom(1)
        e("s_add_u32 s19, s19, 2")
        self.jump_next(0)
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_sqb")
        self.finish_bottom()
        # ---------------- top wave
        e(".p2align 6")
        e("L_sq_top:")
        e(f"v_add_u32 v{g.v_drow}, {WL * row}, v{g.v_aread}")      # diagonal operands: limbs WL, WL+1, ...
        e(f"v_mov_b32 v{g.v_din}, 0")
        self.jump_base("L_sqbase_t")
        e("s_mov_b32 s18, 0")
        e(".p2align 6")
        e("L_sqt:")
        self.set_exp()
        self.sq_row_top(0)
        self.sq_row_top(1)
        e("s_add_u32 s19, s19, 2")
        self.jump_next(1)
        e(f"s_cmp_lt_u32 s19, {WT}")
        e("s_cbranch_scc1 L_sqt")
        e("s_waitcnt lgkmcnt(0)")
        for k in range((WL + 1) // 2, WL):
            self.mad(self.T(2 * k - WL), self.X(k), self.X(k), self.T(2 * k - WL))
        self.finish_top(top_zeroed=True)
        self.lines.extend(self.deferred)
        self.deferred = []

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenP(Gen):
    """Pair kernel for moduli N = p^2 (Paillier's CRT halves): a residue y mod p^2 is kept as two base-p digits
    y~ = y R mod p^2 = a0 + a1 p   (R = 2^(28 H), H = limbs of p; a0, a1 lazily reduced mod p)
    and a product costs five half-width limb products instead of eight:
        a0 b0 = t R - m p            (Montgomery step mod p; m = the quotient digits it computes anyway)
        c0 = t,   c1 = (a0 b1 + a1 b0 + Cadj - m) R^-1 mod p      (one more Montgomery step mod p)
    because a1 b1 p^2 vanishes and the reduction modulo p^2 splits into two reductions modulo p.  Cadj is a multiple of p
    whose limbs all exceed 2^28, so Cadj - m is limb-wise non-negative (no borrows in the lazy accumulators).
    A squaring needs H(H-1)/2 + H + 3 H^2 multiplies instead of (2H)(2H-1)/2 + 2H + 4 H^2: 58 %.
    Same VM contract as vm_asm_74_1 (74-limb values: a0 in limbs 0..H-1, a1 in limbs H..2H-1); SQR and MUL are the pair
    operations; `nmod` points at p (H limbs) followed by Cadj (H limbs), `n0inv` belongs to p.
    Phase 1 (a0 b0 and its reduction) is fully unrolled: static column renaming instead of the register shift, the
    triangular rows of a squaring need no computed jump, and each quotient digit lands in its own register."""

    def __init__(self, H=37):
        Gen.__init__(self, 2 * H, 1)
        assert H % 2 == 1 and 6 * H <= 224
        self.H = H
        self.name = f"vm_asm_{H}_16"
        self.sq_rows = True
        self.sq_self_staged = True # a squaring writes its (doubled) multipliers to the LDS column itself, row by row
        self.nm4_tables = True     # STORET / MULVT: per-number window tables number-major
        self.vM = 2 * H            # quotient digits m_i of phase 1 (phase 2 starts from Cadj_i - m_i)
        self.vA = 3 * H            # new a0 of a MUL (the old one is still an operand of phase 2)
        import os
        self.timing = os.environ.get("PGPU_GEN_TIMING", "0") == "1"        # debug build for tools/wave_timeline.py

    def X0(self, j):
        return self.X(j)

    def X1(self, j):
        return self.X(self.H + j)

    def Pm(self, j):
        return f"s{self.s_N + j}"

    def Cd(self, j):
        return f"s{self.s_N + self.H + j}"

    # ---- phase 1: c0 = a0 * b0 * R^-1 mod p, quotient digits captured -------------------------------------------
    def phase1(self, sq, dest):
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        m = f"v{g.v_m}"
        single = getattr(self, "single_digit", False)      # GenM: the modulus is p itself, there is no phase 2
        fresh = set(range(H))

        def acc(pos, a, b):
            c = pos % H
            if c in fresh:
                fresh.discard(c)
                self.mad(self.T(c), a, b, "0")
            else:
                self.mad(self.T(c), a, b, self.T(c))

        if not sq:
            e(f"ds_read_b32 v{g.v_ain}, v{g.v_aread}")
        regs = [g.v_ain, g.v_ai]
        for i in range(H):
            if sq:
                ai = self.X0(i)
                e(f"v_add_u32 v{g.v_ai}, {ai}, {ai}")
                if not single:
                    e(f"ds_write_b32 v{g.v_awrite}, v{g.v_ai} offset:{i * row}")     # phase 2 multiplies by 2 a0_i
                self.align8()
                acc(2 * i, ai, ai)
                for j in range(i + 1, H):
                    acc(i + j, f"v{g.v_ai}", self.X0(j))
            else:
                cur, nxt = regs[i % 2], regs[(i + 1) % 2]
                e("s_waitcnt lgkmcnt(0)")
                if i + 1 < H:
                    e(f"ds_read_b32 v{nxt}, v{g.v_aread} offset:{(i + 1) * row}")
                self.align8()
                for j in range(H):
                    acc(i + j, f"v{cur}", self.X0(j))
            c0 = i % H
            assert c0 not in fresh
            mi = m if single else f"v{g.vM + i}"     # the quotient digit stays in its own register for phase 2
            e(f"v_mul_lo_u32 {mi}, {self.Tlo(c0)}, s14")
            e(f"v_and_b32 {mi}, {hex(MASK)}, {mi}")
            self.align8()
            for j in range(H):
                acc(i + j, mi, self.Pm(j))
            e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.T(c0)}")
            e(f"v_lshl_add_u64 {self.T((i + 1) % H)}, {self.T((i + 1) % H)}, 0, {self.P(g.v_c)}")
            fresh.add(c0)
        # columns H .. 2H-1 hold c0; the one just retired is position 2H-1... no: position H-1 was retired, i.e. column
        # (H-1) % H is free and position 2H-1 was never opened unless written above
        live = {}
        for pos in range(H, 2 * H):
            c = pos % H
            if c in fresh:
                e(f"v_mov_b64 {self.T(c)}, 0")
                fresh.discard(c)
            live[pos] = c
        # sequential carry into dest[0..H-1]
        M = hex(MASK)
        cpair = self.P(g.v_c)
        first = True
        for k, pos in enumerate(range(H, 2 * H)):
            c = live[pos]
            if not first:
                e(f"v_lshl_add_u64 {self.T(c)}, {self.T(c)}, 0, {cpair}")
            first = False
            e(f"v_and_b32 {dest(k)}, {M}, {self.Tlo(c)}")
            if k < H - 1:
                e(f"v_lshrrev_b64 {cpair}, {LB}, {self.T(c)}")

    # ---- phase 2: c1 = (cross + Cadj - m) R^-1 mod p ---------------------------------------------------------------
    def phase2_row(self, sq, cur, nxt, cur1, nxt1, first, roff=0, bump=None):
        """one row, register-shift style (T(j-1) <- T(j) + m' p_j).  sq: multiplier 2 a0_i times a1; else b0_i times a1
        plus b1_i times a0.  cur/nxt: a_i registers (this row / prefetch); cur1/nxt1: the b1 stream of a MUL."""
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        m = f"v{g.v_m}"
        e("s_waitcnt lgkmcnt(0)")
        e(f"ds_read_b32 v{nxt}, v{g.v_arow} offset:{roff}")
        if not sq:
            e(f"ds_read_b32 v{nxt1}, v{g.v_arow} offset:{H * row + roff}")
        if bump:
            e(f"v_add_u32 v{g.v_arow}, {bump}, v{g.v_arow}")
        a = f"v{cur}"                                   # a squaring reads 2 a0_i: phase 1 stored it doubled
        self.align8()
        for j in range(H):
            if j == H - 1 and not first:
                self.mad(self.T(j), a, self.X1(j), "0")
            else:
                self.mad(self.T(j), a, self.X1(j), self.T(j))
        if not sq:
            for j in range(H):
                self.mad(self.T(j), f"v{cur1}", self.X0(j), self.T(j))
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        self.align8()
        self.mad(self.P(g.v_y0), m, self.Pm(0), self.T(0))
        self.mad(self.T(0), m, self.Pm(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        for j in range(2, H):
            self.mad(self.T(j - 1), m, self.Pm(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")

    def phase2(self, sq, tag):
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        for j in range(H):
            e(f"v_sub_u32 {self.Tlo(j)}, {self.Cd(j)}, v{g.vM + j}")     # Cadj_j - m_j >= 0 limb by limb
            e(f"v_mov_b32 {self.Thi(j)}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        if not sq:
            e(f"ds_read_b32 v{g.v_t2}, v{g.v_arow} offset:{H * row}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        # row 0 keeps the initial value of the top column; then (H-1)/2 iterations of two rows
        self.phase2_row(sq, g.v_ain, g.v_ai, g.v_t2, g.v_t3, True, 0, row)
        e("s_mov_b32 s19, 1")
        e(".p2align 6")
        e(f"L_p2{tag}:")
        self.phase2_row(sq, g.v_ai, g.v_ain, g.v_t3, g.v_t2, False, 0, None)
        self.phase2_row(sq, g.v_ain, g.v_ai, g.v_t2, g.v_t3, False, row, 2 * row)
        e("s_add_u32 s19, s19, 2")
        e(f"s_cmp_lt_u32 s19, {H}")
        e(f"s_cbranch_scc1 L_p2{tag}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_mov_b64 {self.T(H - 1)}, 0")
        M = hex(MASK)
        c = self.P(g.v_c)
        e(f"v_and_b32 {self.X1(0)}, {M}, {self.Tlo(0)}")
        e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
        for j in range(1, H):
            e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            e(f"v_and_b32 {self.X1(j)}, {M}, {self.Tlo(j)}")
            if j < H - 1:
                e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")

    def montsq(self):
        e = self.e
        e("L_montsq:")
        self.phase1(True, self.X0)        # phase 2 reads a0 from the LDS column, so the new a0 can replace the old one now
        self.phase2(True, "s")
        e("s_branch L_next")

    def montmul(self):
        g, e = self, self.e
        H = self.H
        e("L_montmul:")
        self.phase1(False, lambda k: f"v{g.vA + k}")
        self.phase2(False, "m")
        for j in range(H):
            e(f"v_mov_b32 {self.X0(j)}, v{g.vA + j}")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenM(Gen):
    """One-lane kernel for an odd number H of limbs with the product fully unrolled (GenP's phase 1 alone: static column
    renaming instead of the register shift, triangular squaring rows without a computed jump, no LDS traffic in a squaring):
    the ladders modulo the PRIMES of a 2048-bit key (struct_pow_n3, the s ladder).  Same VM contract, slots and operations as
    Gen(H, 1) -- only L_montmul / L_montsq differ; a result is congruent to Gen's modulo p and as lazily reduced."""

    def __init__(self, H=37):
        Gen.__init__(self, H, 1)
        assert H % 2 == 1 and self.n_sgpr
        self.H = H
        self.sq_rows = True
        self.sq_self_staged = True     # a squaring reads its operand from the registers only
        self.single_digit = True
        self.nm4_tables = True

    X0 = GenP.X0
    Pm = GenP.Pm
    phase1 = GenP.phase1

    def X1(self, j):
        raise AssertionError("single digit")

    def montsq(self):
        self.e("L_montsq:")
        self.phase1(True, self.X0)
        self.e("s_branch L_next")

    def montmul(self):
        self.e("L_montmul:")
        self.phase1(False, self.X0)      # (b = X is last read by the final row's products; the carry pass comes after them)
        self.e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenP2(GenP):
    """One-lane pair kernel for primes too wide for GenP's register map (H = 55 limbs: the CRT halves p^2, q^2 of 3072-bit
    keys).  Same arithmetic and the same digits as GenP -- a squaring is 3.5 H^2 multiplies, a product 5 H^2, where the
    two-lane kernel (GenQ) issues 4 H^2 and 8 H^2 -- but 5 H registers of operands and accumulators plus H quotient digits
    do not fit 256 VGPRs (two waves per SIMD) at H = 55, and neither does GenP's LDS staging of both multiplier streams
    (2H + 1 words per lane: 113 KB per block).  What moves:
      * the quotient digits m_i of phase 1 go to the lane's LDS column (one ds_write per row, no VALU slot) and come back as
        the seeds Cadj_j - m_j of phase 2; Cadj sits in an LDS table the block fills once (p alone takes 55 SGPRs);
      * a squaring stores its doubled multipliers 2 a0_k over the same LDS words at the phase boundary, where the new a0
        replaces the old one -- H + 1 words per lane in all: 56.5 KB per block, two blocks per CU;
      * a product streams its multiplier digits b0_i, b1_i straight from memory (slot, constant) through two rings of five
        registers, four rows ahead of their use (vmcnt-counted), and parks the new a0 in the LDS column until phase 2,
        which still multiplies by the old one, has finished.  Both phases of a product are loops of five-row bodies.
    VGPRs: T 2H | a0 H | a1 H | 29 others = 249."""

    RB = 5   # rows per loop body of a product = registers per multiplier ring
    PD = 4   # rows between a multiplier's load and its use

    def __init__(self, H=55):
        assert H % 2 == 1 and H % self.RB == 0 and H > 2 * self.RB
        self.H = H
        self.WL = self.WT = 2 * H
        self.K = 1
        self.NPB = BLOCK
        self.depth = 8
        self.lines, self.deferred = [], []
        self.name = f"vm_asm_{H}_16"
        self.sq_rows, self.sq_rows_k, self.sq_self_staged = True, False, True
        self.n_sgpr, self.n_vreg, self.flush = True, False, False
        self.vX = 2 * H
        e = 4 * H
        for nm in ["ai", "ain", "m", "t1", "t2", "t3", "t4", "goff", "aread", "arow", "mrow", "zero", "vb"]:
            setattr(self, "v_" + nm, e)
            e += 1
        self.v_awrite, self.v_addr, self.v_koff = self.v_aread, self.v_arow, self.v_zero
        self.ring0 = list(range(e, e + self.RB))
        e += self.RB
        self.ring1 = list(range(e, e + self.RB))
        e += self.RB
        e = (e + 1) // 2 * 2
        self.v_y0 = e
        e += 2
        self.v_c = self.v_p1 = e
        e += 2
        self.n_vgpr = e
        assert e <= 256, f"VGPR budget exceeded: {e}"
        self.s_N = 20
        assert self.s_N + H <= 76
        self.s_b0, self.s_b1, self.s_bstride, self.s_bstart = 76, 78, 80, 82
        self.s_sbase = 94
        self.s_t0, self.s_t1 = 96, 97
        self.n_sgpr_count = 102
        self.lds_n = 0
        self.lds_a = (H * 4 + 255) // 256 * 256            # Cadj table first, then the lanes' columns [H + 1][256]
        self.lds_bytes = self.lds_a + (H + 1) * BLOCK * 4
        self.row = BLOCK * 4
        self.nm4_tables = False    # (its own dispatcher: 4-bit limb-major windows only)

    # ---------------------------------------------------------------------------------------------
    def prologue(self):
        g, e, H = self, self.e, self.H
        e('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")    # prog, nmod, consts, mem
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")  # digits, n0inv, nb
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")                  # byte stride between limb rows
        e(f"s_lshl_b32 s{g.s_t0}, s2, 8")
        e(f"v_add_lshl_u32 v{g.v_goff}, s{g.s_t0}, v0, 2")      # (blk * 256 + lane) * 4
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v0")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mov_b32 v{g.v_zero}, 0")
        off, s, rem = 0, self.s_N, H                            # p -> SGPRs
        while rem > 0:
            for cnt in (16, 8, 4, 2, 1):
                align = 4 if cnt >= 4 else cnt
                if cnt <= rem and s % align == 0:
                    if cnt == 1:
                        e(f"s_load_dword s{s}, s[6:7], {hex(off)}")
                    else:
                        e(f"s_load_dwordx{cnt} s[{s}:{s + cnt - 1}], s[6:7], {hex(off)}")
                    off += 4 * cnt
                    s += cnt
                    rem -= cnt
                    break
            else:
                raise RuntimeError("cannot tile the modulus into SGPR loads")
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")                    # Cadj -> LDS words [0, H)
        e(f"v_cmp_gt_u32 vcc, {H}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        e(f"global_load_dword v{g.v_t1}, v{g.v_t3}, s[6:7] offset:{H * 4}")
        e("s_waitcnt vmcnt(0)")
        e(f"ds_write_b32 v{g.v_t3}, v{g.v_t1}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(2 * H):
            e(f"v_mov_b32 {self.X(j)}, 0")

    def dispatcher(self):
        g, e, H = self, self.e, self.H
        Xs = [self.X(j) for j in range(2 * H)]
        St = [f"v{j}" for j in range(2 * H)]
        M = hex(MASK)
        e("L_next:")
        e("s_load_dwordx2 s[16:17], s[4:5], 0x0")
        e("s_add_u32 s4, s4, 8")
        e("s_addc_u32 s5, s5, 0")
        e("s_waitcnt lgkmcnt(0)")
        self.fair_share()
        e("s_and_b32 s18, s16, 0xff")
        for nm in ("SQR", "MUL", "MULC", "MULV", "LOAD", "STORE", "LOADC", "ADD"):
            e(f"s_cmp_eq_u32 s18, {OPS[nm]}")
            e(f"s_cbranch_scc1 L_{nm.lower()}")
        self.end_of_program()  # END (per-number windows: 4-bit VM_MULV only; the host refuses other table opcodes for this kernel)

        e("L_load:")
        self.slot_base()
        self.load_slot_into(Xs)
        e("s_branch L_next")
        e("L_loadc:")
        self.const_base()
        self.load_const_into(Xs)
        e("s_branch L_next")
        e("L_store:")
        self.slot_base()
        e(f"v_mov_b32 v{g.v_addr}, v{g.v_goff}")
        for j in range(2 * H):
            e(f"global_store_dword v{g.v_addr}, {Xs[j]}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            if j != 2 * H - 1:
                e(f"v_add_u32 v{g.v_addr}, s3, v{g.v_addr}")
        e("s_waitcnt vmcnt(0)")
        e("s_branch L_next")
        e("L_add:")
        # digit-wise sum, carried at once (GenP leaves it lazy; at H = 55 the accumulators of the product that follows have
        # no room for 29-bit limbs: 3 H products of 2^57).  The digits' values are what the next product sees, not the limbs.
        self.slot_base()
        self.load_slot_into(St)
        for j in range(2 * H):
            e(f"v_add_u32 {Xs[j]}, {Xs[j]}, {St[j]}")
        for d in (0, H):
            for j in range(H - 1):
                e(f"v_lshrrev_b32 v{g.v_t1}, {LB}, {Xs[d + j]}")
                e(f"v_and_b32 {Xs[d + j]}, {M}, {Xs[d + j]}")
                e(f"v_add_u32 {Xs[d + j + 1]}, {Xs[d + j + 1]}, v{g.v_t1}")
        e("s_branch L_next")
        e("L_mul:")
        self.slot_base()
        e(f"s_mov_b64 s[{g.s_bstart}:{g.s_bstart + 1}], s[{g.s_sbase}:{g.s_sbase + 1}]")
        e(f"s_mov_b32 s{g.s_bstride}, s3")
        e(f"v_mov_b32 v{g.v_vb}, v{g.v_goff}")
        e("s_branch L_montmul")
        e("L_mulc:")
        self.const_base()
        e(f"s_mov_b64 s[{g.s_bstart}:{g.s_bstart + 1}], s[{g.s_sbase}:{g.s_sbase + 1}]")
        e(f"s_mov_b32 s{g.s_bstride}, 4")
        e(f"v_mov_b32 v{g.v_vb}, 0")
        e("s_branch L_montmul")
        e("L_mulv:")
        # per-number table index: 4-bit window `arg` of this number's own exponent (7 windows per 28-bit limb of `digits`);
        # the operand streams from slot aux + digit -- the same multiplier rings as MUL, with a per-lane address
        e(f"s_mul_hi_u32 s{g.s_t1}, s17, {((1 << 32) + 6) // 7}")            # q = arg / 7
        e(f"s_mul_i32 s98, s{g.s_t1}, 7")
        e("s_sub_u32 s98, s17, s98")
        e("s_lshl_b32 s98, s98, 2")                                          # shift = 4 (arg % 7)
        e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s{g.s_t1}, s3")                    # digits + q * nb*4
        e(f"s_mul_i32 s{g.s_sbase}, s{g.s_t1}, s3")
        e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s12")
        e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s13")
        e(f"global_load_dword v{g.v_t3}, v{g.v_goff}, s[{g.s_sbase}:{g.s_sbase + 1}]")
        e("s_waitcnt vmcnt(0)")
        e(f"v_lshrrev_b32 v{g.v_t3}, s98, v{g.v_t3}")
        e(f"v_and_b32 v{g.v_t3}, 15, v{g.v_t3}")                             # digit
        e(f"s_mul_i32 s{g.s_t0}, s3, {2 * H}")                                # slot stride in bytes (< 2^32: host)
        e(f"v_mul_lo_u32 v{g.v_t3}, v{g.v_t3}, s{g.s_t0}")                    # digit * stride (the host keeps 17 slots below 2^32)
        e(f"v_add_u32 v{g.v_vb}, v{g.v_t3}, v{g.v_goff}")
        e("s_bfe_u32 s17, s16, 0x160008")                                    # aux = first table slot
        self.slot_base()
        e(f"s_mov_b64 s[{g.s_bstart}:{g.s_bstart + 1}], s[{g.s_sbase}:{g.s_sbase + 1}]")
        e(f"s_mov_b32 s{g.s_bstride}, s3")
        e("s_branch L_montmul")
        e("L_sqr:")
        e("s_branch L_montsq")

    # ---- the phase boundary: c0 leaves the accumulators, the seeds Cadj_k - m_k enter them ---------------------------
    def boundary(self, sq):
        """Columns k = 0..H-1 hold c0 (position H + k).  Carry them out one by one; a freed column at once receives m_k (from
        the lane's LDS column) in its low and Cadj_k (LDS table) in its high register.  Squaring: the LDS word takes 2 a0_k
        (the multiplier of phase-2 row k) and the new a0_k replaces the old one; product: the LDS word takes the new a0_k."""
        g, e, H = self, self.e, self.H
        M = hex(MASK)
        cpair = self.P(g.v_c)
        tmp = [g.v_t1, g.v_t2]
        for k in range(H):
            t = f"v{tmp[k % 2]}"
            if k:
                e(f"v_lshl_add_u64 {self.T(k)}, {self.T(k)}, 0, {cpair}")
            if sq:
                e(f"v_add_u32 {t}, {self.X0(k)}, {self.X0(k)}")
                e(f"v_and_b32 {self.X0(k)}, {M}, {self.Tlo(k)}")
            else:
                e(f"v_and_b32 {t}, {M}, {self.Tlo(k)}")
            if k < H - 1:
                e(f"v_lshrrev_b64 {cpair}, {LB}, {self.T(k)}")
            e(f"ds_read_b32 {self.Tlo(k)}, v{g.v_aread} offset:{k * self.row}")
            e(f"ds_read_b32 {self.Thi(k)}, v{g.v_zero} offset:{4 * k}")
            e(f"ds_write_b32 v{g.v_aread}, {t} offset:{k * self.row}")

    def seeds(self, fold_top):
        g, e, H = self, self.e, self.H
        e("s_waitcnt lgkmcnt(0)")
        for j in range(H):
            e(f"v_sub_u32 {self.Tlo(j)}, {self.Thi(j)}, {self.Tlo(j)}")     # Cadj_j - m_j >= 0 limb by limb
            e(f"v_mov_b32 {self.Thi(j)}, 0")
        if fold_top:
            # the rows of a product re-create the top column from zero, every one of them (no peeled first row): its seed
            # moves into the column below, 28 bits up (< 2^57: the accumulators have the room)
            e(f"v_lshlrev_b64 {self.P(g.v_y0)}, {LB}, {self.T(H - 1)}")
            e(f"v_lshl_add_u64 {self.T(H - 2)}, {self.T(H - 2)}, 0, {self.P(g.v_y0)}")

    # ---- squaring -----------------------------------------------------------------------------------------------
    def phase1_sq(self):
        g, e, H = self, self.e, self.H
        m = f"v{g.v_m}"
        single = getattr(self, "single_digit", False)      # GenM: the modulus is p itself, there is no phase 2
        fresh = set(range(H))

        def acc(pos, a, b):
            c = pos % H
            if c in fresh:
                fresh.discard(c)
                self.mad(self.T(c), a, b, "0")
            else:
                self.mad(self.T(c), a, b, self.T(c))

        for i in range(H):
            ai = self.X0(i)
            e(f"v_add_u32 v{g.v_ai}, {ai}, {ai}")
            self.align8()
            acc(2 * i, ai, ai)
            for j in range(i + 1, H):
                acc(i + j, f"v{g.v_ai}", self.X0(j))
            c0 = i % H
            assert c0 not in fresh
            e(f"v_mul_lo_u32 {m}, {self.Tlo(c0)}, s14")
            e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
            e(f"ds_write_b32 v{g.v_aread}, {m} offset:{i * self.row}")
            self.align8()
            for j in range(H):
                acc(i + j, m, self.Pm(j))
            e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.T(c0)}")
            e(f"v_lshl_add_u64 {self.T((i + 1) % H)}, {self.T((i + 1) % H)}, 0, {self.P(g.v_c)}")
            fresh.add(c0)
        for pos in range(H, 2 * H):
            c = pos % H
            assert c == pos - H
            if c in fresh:
                e(f"v_mov_b64 {self.T(c)}, 0")
                fresh.discard(c)

    def carry_into_x1(self):
        g, e, H = self, self.e, self.H
        M = hex(MASK)
        c = self.P(g.v_c)
        e(f"v_mov_b64 {self.T(H - 1)}, 0")
        e(f"v_and_b32 {self.X1(0)}, {M}, {self.Tlo(0)}")
        e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
        for j in range(1, H):
            e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            e(f"v_and_b32 {self.X1(j)}, {M}, {self.Tlo(j)}")
            if j < H - 1:
                e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")

    def montsq(self):
        g, e, H = self, self.e, self.H
        row = self.row
        e("L_montsq:")
        self.phase1_sq()
        self.boundary(True)
        self.seeds(False)
        # phase 2 as in GenP: multipliers 2 a0_i from the LDS column, row 0 keeps the seed of the top column
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        e(f"ds_read_b32 v{g.v_ain}, v{g.v_arow}")
        e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
        self.phase2_row(True, g.v_ain, g.v_ai, g.v_t2, g.v_t3, True, 0, row)
        e("s_mov_b32 s19, 1")
        e(".p2align 6")
        e("L_p2s:")
        self.phase2_row(True, g.v_ai, g.v_ain, g.v_t3, g.v_t2, False, 0, None)
        self.phase2_row(True, g.v_ain, g.v_ai, g.v_t2, g.v_t3, False, row, 2 * row)
        e("s_add_u32 s19, s19, 2")
        e(f"s_cmp_lt_u32 s19, {H}")
        e("s_cbranch_scc1 L_p2s")
        e("s_waitcnt lgkmcnt(0)")
        self.carry_into_x1()
        e("s_branch L_next")

    # ---- product ------------------------------------------------------------------------------------------------
    def issue_loads(self, slot, both):
        """multiplier digits of the next row in line -> ring registers `slot`; the row pointers move on"""
        g, e = self, self.e
        e(f"global_load_dword v{g.ring0[slot]}, v{g.v_vb}, s[{g.s_b0}:{g.s_b0 + 1}]")
        e(f"s_add_u32 s{g.s_b0}, s{g.s_b0}, s{g.s_bstride}")
        e(f"s_addc_u32 s{g.s_b0 + 1}, s{g.s_b0 + 1}, 0")
        if both:
            e(f"global_load_dword v{g.ring1[slot]}, v{g.v_vb}, s[{g.s_b1}:{g.s_b1 + 1}]")
            e(f"s_add_u32 s{g.s_b1}, s{g.s_b1}, s{g.s_bstride}")
            e(f"s_addc_u32 s{g.s_b1 + 1}, s{g.s_b1 + 1}, 0")

    def mul_row(self, r, phase, rows_left):
        """row r of a five-row body.  phase 1: T += b0_i a0, quotient digit to LDS, T = (T + m p) >> 28.  phase 2:
        T += b0_i a1 + b1_i a0, the same reduction.  rows_left: rows after this one in the product's phase (tail body), or
        None inside the loop: decides the vmcnt to wait for and whether row i + PD is still to be fetched."""
        g, e, H = self, self.e, self.H
        per = 2 if phase == 2 else 1
        newer = self.PD - 1 if rows_left is None else min(self.PD - 1, rows_left)
        e(f"s_waitcnt vmcnt({per * newer})")
        if rows_left is None or rows_left >= self.PD:
            self.issue_loads((r + self.PD) % self.RB, phase == 2)
        b0 = f"v{g.ring0[r]}"
        m = f"v{g.v_m}"
        self.align8()
        if phase == 1:
            for j in range(H):
                self.mad(self.T(j), b0, self.X0(j), "0" if j == H - 1 else self.T(j))
        else:
            b1 = f"v{g.ring1[r]}"
            for j in range(H):
                self.mad(self.T(j), b0, self.X1(j), "0" if j == H - 1 else self.T(j))
            for j in range(H):
                self.mad(self.T(j), b1, self.X0(j), self.T(j))
        e(f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14")
        e(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        if phase == 1:
            e(f"ds_write_b32 v{g.v_mrow}, {m} offset:{r * self.row}")
        self.align8()
        self.mad(self.P(g.v_y0), m, self.Pm(0), self.T(0))
        self.mad(self.T(0), m, self.Pm(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        for j in range(2, H):
            self.mad(self.T(j - 1), m, self.Pm(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")

    def mul_phase(self, phase):
        g, e, H, RB = self, self.e, self.H, self.RB
        e(f"s_mov_b64 s[{g.s_b0}:{g.s_b0 + 1}], s[{g.s_bstart}:{g.s_bstart + 1}]")
        if phase == 2:
            e(f"s_mul_i32 s{g.s_t0}, s{g.s_bstride}, {H}")
            e(f"s_add_u32 s{g.s_b1}, s{g.s_bstart}, s{g.s_t0}")
            e(f"s_addc_u32 s{g.s_b1 + 1}, s{g.s_bstart + 1}, 0")
        for r in range(self.PD):
            self.issue_loads(r, phase == 2)
        if phase == 1:
            for j in range(H - 1):
                e(f"v_mov_b64 {self.T(j)}, 0")
            e(f"v_mov_b32 v{g.v_mrow}, v{g.v_aread}")
        else:
            self.seeds(True)
        e("s_mov_b32 s19, 0")
        e(".p2align 6")
        e(f"L_m{phase}:")
        for r in range(RB):
            self.mul_row(r, phase, None)
        if phase == 1:
            e(f"v_add_u32 v{g.v_mrow}, {RB * self.row}, v{g.v_mrow}")
        e(f"s_add_u32 s19, s19, {RB}")
        e(f"s_cmp_lt_u32 s19, {H - RB}")
        e(f"s_cbranch_scc1 L_m{phase}")
        for r in range(RB):
            self.mul_row(r, phase, RB - 1 - r)

    def montmul(self):
        g, e, H = self, self.e, self.H
        e("L_montmul:")
        self.mul_phase(1)
        e(f"v_mov_b64 {self.T(H - 1)}, 0")
        self.boundary(False)
        self.mul_phase(2)
        self.carry_into_x1()
        for k in range(H):
            e(f"ds_read_b32 {self.X0(k)}, v{g.v_aread} offset:{k * self.row}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class LaneRows:
    """Rows of the kernels whose lanes own whole digits (GenQ: two lanes of a number modulo n^2; GenQ3: a quad modulo n^3): the
    single-lane Montgomery row modulo n with the modulus in SGPRs, the quotient links between the lanes interleaved with the
    multiplies, multipliers read (and, in a squaring, doubled) two rows at a time, four-row loop bodies.  hops(link) of the
    kernel: the (DPP control, lane mask register) of every link of a row."""

    def row(self, a, link2, first=False, a2=None):
        """one Montgomery row modulo n in every active lane on the multiplier a (a2: the second stream of a one-pass product).
        Hop 1: lane 1 takes -m(lane 0) into column 0 (its C1 limb is in the accumulator since the pass began); hop 2 (link2):
        lane 2 takes -m(lane 1).  first: row 0 of a pass (every accumulator, the top one too, still holds its initial constant).
        Column 0 is complete after the FIRST multiply of pass A (only a * x_0 lands in it), so the quotient digit and the two
        link hops (each a chain of dependent instructions: multiply, mask + DPP move in one instruction, signed multiply-add)
        are started right away and their steps are spread between the remaining multiplies of pass A: no s_nop for the DPP
        hazard, and at one wave per SIMD -- a 16 384-number batch -- nothing waits on a result that is still in flight."""
        g, e = self, self.e
        H = self.H
        N = lambda j: f"s{g.s_N + j}"
        m = f"v{g.v_m}"
        chain = [f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        for ctrl, mask in self.hops(link2):
            chain += [f"v_and_b32_dpp v{g.v_d}, {m}, v{g.v_mask28} {ctrl} row_mask:0xf bank_mask:0xf",
                      f"v_mad_i64_i32 {self.T(0)}, vcc, v{g.v_d}, v{mask}, {self.T(0)}",     # T0 -= m of the lane below (mask: -1 / 0)
                      f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        chain.append(f"v_and_b32 {m}, {hex(MASK)}, {m}")
        addend = lambda j: "0" if (j == H - 1 and not first) else self.T(j)
        muls = []
        for j in range(H):
            muls.append((self.T(j), f"v{a}", self.X(j), addend(j)))
            if a2 is not None:
                muls.append((self.T(j), f"v{a2}", f"v{g.vY + j}", self.T(j)))
        head = 1 if a2 is None else 2                      # multiplies that complete column 0
        gap = min(4, (len(muls) - head) // len(chain))     # multiplies between two steps of the chain (>= 2 covers the DPP hazard)
        assert gap >= 2, "pass A is too short to hide the link chain"
        self.align8()
        k = 0
        for _ in range(head):
            self.mad(*muls[k])
            k += 1
        for step in chain:
            for _ in range(gap):
                self.mad(*muls[k])
                k += 1
            e(step)
        while k < len(muls):
            self.mad(*muls[k])
            k += 1
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        for j in range(2, H):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")

    def read_pair(self, dst, ptr, first_row):
        """multipliers of rows first_row, first_row + 1 past the row pointer -> the register pair dst"""
        u = self.NPB * 4 // 256                            # a row in units of 64 dwords
        assert u * 256 == self.NPB * 4 and (first_row + 1) * u <= 255
        self.e(f"ds_read2st64_b32 v[{dst}:{dst + 1}], v{ptr} offset0:{first_row * u} offset1:{(first_row + 1) * u}")

    def row_pair(self, cur, nxt, link2, use_sh, ahead, bump, first=False, cur2=None, nxt2=None, loop=None):
        """two rows on the multiplier pair(s) cur; the pair(s) `ahead` rows past the pointers are fetched into nxt; bump: rows the
        pointers move on afterwards; loop: 'count' moves the loop counter here, (limit) compares it here -- the 4-byte scalar
        instructions sit in pairs so that the multiply streams stay on 8-byte boundaries without padding (a padding s_nop is an
        issue slot like any other at one wave per SIMD)."""
        g, e = self, self.e
        rowb = self.NPB * 4
        e("s_waitcnt lgkmcnt(0)")
        if loop == "count":
            e("s_add_u32 s19, s19, 1")
        elif loop is not None:
            e(f"s_cmp_lt_u32 s19, {loop}")
        else:
            e("s_nop 0")
        self.read_pair(nxt, g.v_arow, ahead)
        if cur2 is not None:
            self.read_pair(nxt2, g.v_arow2, ahead)
        if bump:
            e(f"v_add_u32 v{g.v_arow}, {bump * rowb}, v{g.v_arow}")
            if cur2 is not None:
                e(f"v_lshl_add_u32 v{g.v_arow2}, v{g.v_bump2}, {1 if bump == 2 else 2}, v{g.v_arow2}")
        if use_sh:
            e(f"v_lshlrev_b64 {self.P(cur)}, v{g.v_sh}, {self.P(cur)}")     # both multipliers at once: 28-bit limbs stay in their words
        self.row(cur, link2, first, cur2)
        self.row(cur + 1, link2, False, None if cur2 is None else cur2 + 1)

    def rows(self, tag, link2, use_sh, two_streams=False):
        """the H rows of a pass: the accumulators hold their initial values, the row pointer(s) are at row 0 of the stream(s)"""
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        s2 = lambda c2, n2: dict(cur2=c2, nxt2=n2) if two_streams else {}
        A, B = (g.v_pa, getattr(g, "v_pa2", None)), (g.v_pb, getattr(g, "v_pb2", None))
        rows = H
        first = True
        if H % 2:
            # an odd row count: row 0 on its own (its multiplier goes to the second register of the idle pair)
            e(f"ds_read_b32 v{B[0] + 1}, v{g.v_arow}")
            if two_streams:
                e(f"ds_read_b32 v{B[1] + 1}, v{g.v_arow2}")
            e(f"v_add_u32 v{g.v_arow}, {row}, v{g.v_arow}")
            if two_streams:
                e(f"v_add_u32 v{g.v_arow2}, v{g.v_bump2}, v{g.v_arow2}")
            self.read_pair(A[0], g.v_arow, 0)
            if two_streams:
                self.read_pair(A[1], g.v_arow2, 0)
            e("s_waitcnt lgkmcnt(0)")
            if use_sh:
                e(f"v_lshlrev_b32 v{B[0] + 1}, v{g.v_sh}, v{B[0] + 1}")
            self.row(B[0] + 1, link2, True, (B[1] + 1) if two_streams else None)
            rows -= 1
            first = False
        else:
            self.read_pair(A[0], g.v_arow, 0)
            if two_streams:
                self.read_pair(A[1], g.v_arow2, 0)
        pairs = rows // 2
        while first or pairs % 2:
            self.row_pair(A[0], B[0], link2, use_sh, 2, 2, first=first, **s2(A[1], B[1]))
            A, B = B, A
            pairs -= 1
            first = False
        assert pairs > 0 and pairs % 2 == 0
        e("s_mov_b32 s19, 0")
        e(".p2align 6")
        e(f"L_q{tag}:")
        self.row_pair(A[0], B[0], link2, use_sh, 2, 0, loop="count", **s2(A[1], B[1]))
        self.row_pair(B[0], A[0], link2, use_sh, 4, 4, loop=pairs // 2, **s2(B[1], A[1]))
        e(f"s_cbranch_scc1 L_q{tag}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_mov_b64 {self.T(H - 1)}, 0")


class GenQ(LaneRows, Gen):
    """In-wave pair kernel for moduli N = n^2 with n PUBLIC (Encrypt, ConstMult, PartialDecrypt, proofs): the residue
    y R mod n^2 = a0 + a1 n (R = 2^(28 H), H = limbs of n) lives in two neighbouring lanes -- lane 0 holds a0, lane 1
    holds a1 -- and both lanes run the single-lane Montgomery row modulo n (n in SGPRs, the same for both lanes):
        lane 0:  t  = a0 b0 R^-1                     quotient digits m_i
        lane 1:  c1 = (a1 b0 [+ ...] + Cadj - m) R^-1     m_i arrives by DPP in the row that needs it (column 0 of row i)
    A squaring is ONE pass (lane 0: a0 a0, lane 1: 2 a0 a1): 2 H^2 multiplies per lane instead of 3 H^2 for the
    symmetric 2H-limb squaring split over two waves.  A product needs a second pass for a0 b1 (lane 0; its result
    crosses to lane 1 through the LDS rows the multiplier b1 occupied): 4 H^2 per lane, what the 2H-limb product costs.
    Slot layout = the two-lane layout of every other K = 2 shape: limbs 0..H-1 = a0, H..2H-1 = a1.
    `nmod` points at n (H limbs) followed by Cadj (H limbs, streamed one word per row by scalar loads)."""

    def __init__(self, H=74):
        Gen.__init__(self, H, 2)
        assert 3 * H + 3 <= 255
        self.H = H
        self.name = f"vm_asm_{H}_32"
        self.nm4_tables = True     # STORET / MULVT / MULVT5: per-number window tables number-major
        self.n_sgpr = True
        self.n_vreg = False
        self.flush = False
        self.sq_rows = True
        self.sq_rows_k = False
        self.lds_a = 0
        self.lds_bytes = (self.WT + 1) * self.NPB * 4
        # VGPR map of the single-lane shape plus the lane link
        self.vX = 2 * H
        e = 3 * H
        for nm in ["m", "t1", "sh", "l1mask", "mask28"]:
            setattr(self, "v_" + nm, e)
            e += 1
        e = (e + 1) // 2 * 2
        self.v_pa, self.v_pb = e, e + 2      # multiplier pairs: two rows are read at a time
        e += 4
        self.v_y0 = e
        e += 2
        self.v_d = e          # pair (adjustment, 0)
        e += 2
        for nm in ["goff", "aread", "awrite", "arow", "nbase", "isfirst", "notlast", "t2", "t3", "t4", "koff"]:
            setattr(self, "v_" + nm, e)
            e += 1
        self.v_addr = self.v_arow
        e = (e + 1) // 2 * 2
        self.v_p0 = e
        e += 2
        self.v_p1 = e
        self.v_c = e
        e += 2
        self.v_caddr = e
        e += 1
        self.n_vgpr = e
        assert e <= 256, e
        # constants table in LDS: [H][2 lanes] zero-extended limbs (0 | Cadj_j): the accumulators of a linked pass START there
        # (limb i of Cadj is the initial value of column i of digit one), as in GenQ3 -- no per-row scalar load / subtract / mask
        self.lds_c = 0                                   # (in front of the a columns: LDS instruction offsets are 16 bits)
        self.lds_a = H * 16
        self.lds_bytes = self.lds_a + (self.WT + 1) * self.NPB * 4

    def prologue(self):
        # the K = 2 lane mapping of the base class, with the modulus (H limbs, shared by both lanes) in SGPRs
        g, e = self, self.e
        H, NPB = self.H, self.NPB
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 1, v0")                 # k
        e(f"v_lshrrev_b32 v{g.v_t2}, 1, v0")             # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")  # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {H}")
        e(f"v_mul_lo_u32 v{g.v_t4}, v{g.v_t1}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_t4}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_t4}, {H * NPB * 4}, v{g.v_t1}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_t4}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {H * 4}, v{g.v_t1}")
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_not_b32 v{g.v_l1mask}, v{g.v_isfirst}")
        e(f"v_mov_b32 v{g.v_mask28}, {hex(MASK)}")
        off, s, rem = 0, self.s_N, H
        while rem > 0:
            for cnt in (16, 8, 4, 2, 1):
                align = 4 if cnt >= 4 else cnt
                if cnt <= rem and s % align == 0:
                    if cnt == 1:
                        e(f"s_load_dword s{s}, s[6:7], {hex(off)}")
                    else:
                        e(f"s_load_dwordx{cnt} s[{s}:{s + cnt - 1}], s[6:7], {hex(off)}")
                    off += 4 * cnt
                    s += cnt
                    rem -= cnt
                    break
            else:
                raise RuntimeError("cannot tile the modulus into SGPR loads")
        e("s_waitcnt lgkmcnt(0)")
        # constants table: thread t < H writes row t = (0 | Cadj_t), zero-extended to 64 bits
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")   # this lane's column of the table: k * 8
        e(f"v_cmp_gt_u32 vcc, {H}, v0")
        e("s_and_saveexec_b64 s[96:97], vcc")
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{4 * H}")
        e(f"v_mov_b32 v{g.v_p1 + 1}, 0")
        e(f"v_mov_b32 v{g.v_p0}, 0")
        e(f"v_mov_b32 v{g.v_p0 + 1}, 0")
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v{g.v_t3}")      # t * 16
        e("s_waitcnt vmcnt(0)")
        e(f"ds_write_b64 v{g.v_t3}, {self.P(g.v_p0)} offset:{self.lds_c}")
        e(f"ds_write_b64 v{g.v_t3}, {self.P(g.v_p1)} offset:{self.lds_c + 8}")
        e("s_mov_b64 exec, s[96:97]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(H):
            e(f"v_mov_b32 {self.X(j)}, 0")

    def hops(self, link):
        return [("quad_perm:[0,0,2,2]", self.v_l1mask)] if link else []     # lane 1: T0 -= m of lane 0

    def passes(self, tag, aoff, link, use_sh):
        """H rows: T <- (multiplier stream at byte offset aoff of the a column) * X * R^-1 (+ the lane link: lane 1 takes -m_i of
        lane 0 into column 0 of row i; its Cadj limb has been in the accumulator since the pass began)"""
        g, e = self, self.e
        H = self.H
        if link:
            for j in range(H):                                   # accumulators <- (0 | Cadj_j) by lane
                e(f"ds_read_b64 {self.T(j)}, v{g.v_caddr} offset:{self.lds_c + 16 * j}")
        else:
            for j in range(H):
                e(f"v_mov_b64 {self.T(j)}, 0")
        if aoff:
            e(f"v_add_u32 v{g.v_arow}, {aoff}, v{g.v_aread}")
        else:
            e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        self.rows(tag, link, use_sh)

    def normalize_to(self, dest):
        """sequential carry (every lane owns a whole H-limb digit); dest(j, reg_lo) emits the store of limb j"""
        g, e = self, self.e
        H = self.H
        M = hex(MASK)
        c = self.P(g.v_c)
        for j in range(H):
            if j:
                e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            dest(j)
            if j < H - 1:
                e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")

    def montsq(self):
        g, e = self, self.e
        M = hex(MASK)
        e("L_montsq:")
        e(f"v_and_b32 v{g.v_sh}, 1, v{g.v_l1mask}")           # lane 1 doubles its multiplier: 2 a0 a1
        self.passes("s", 0, True, True)
        self.normalize_to(lambda j: e(f"v_and_b32 {self.X(j)}, {M}, {self.Tlo(j)}"))
        e("s_branch L_next")

    def montmul(self):
        g, e = self, self.e
        H = self.H
        M = hex(MASK)
        row = self.NPB * 4
        e("L_montmul:")
        # pass 1: lane 0: r2 = a0 b1 R^-1 (multiplier rows H..2H-1); lane 1 runs along, its result is dropped
        self.passes("m1", H * row, False, False)
        e(f"v_add_u32 v{g.v_t4}, {H * row}, v{g.v_aread}")       # rows H.. of the a column (offsets must stay below 64 KB)
        e("s_mov_b64 s[96:97], exec")
        e("s_mov_b32 s98, 0x55555555")
        e("s_mov_b32 exec_lo, s98")
        e("s_mov_b32 exec_hi, s98")

        def to_lds(j):
            e(f"v_and_b32 v{g.v_t1}, {M}, {self.Tlo(j)}")
            e(f"ds_write_b32 v{g.v_t4}, v{g.v_t1} offset:{j * row}")
        self.normalize_to(to_lds)
        e("s_mov_b64 exec, s[96:97]")
        # pass 2: lane 0: t = a0 b0 R^-1; lane 1: r1 = (a1 b0 + Cadj - m) R^-1
        self.passes("m2", 0, True, False)
        self.normalize_to(lambda j: e(f"v_and_b32 {self.X(j)}, {M}, {self.Tlo(j)}"))
        # lane 1: c1 = r1 + r2, limbs made canonical again (the top limb keeps the excess: c1 < 4n).  A lazy 29-bit limb would
        # be harmless for typical values, but a doubled multiplier times a 29-bit multiplicand over 74 rows can exceed the 64-bit
        # column in the worst case; 28-bit limbs keep every column below 223 product units.
        e("s_mov_b32 s98, 0xaaaaaaaa")
        e("s_mov_b32 exec_lo, s98")
        e("s_mov_b32 exec_hi, s98")
        St = [f"v{j}" for j in range(H)]
        for j in range(H):
            e(f"ds_read_b32 {St[j]}, v{g.v_t4} offset:{j * row}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(H):
            if j:
                e(f"v_add3_u32 {self.X(j)}, {self.X(j)}, {St[j]}, v{g.v_t1}")
            else:
                e(f"v_add_u32 {self.X(j)}, {self.X(j)}, {St[j]}")
            if j < H - 1:
                e(f"v_lshrrev_b32 v{g.v_t1}, {LB}, {self.X(j)}")
                e(f"v_and_b32 {self.X(j)}, {M}, {self.X(j)}")
        e("s_mov_b64 exec, s[96:97]")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenQ4(Gen):
    """The two-lane pair kernel with every digit sliced over two lanes: 4 lanes per number (lanes 0,1 = the two 37-limb
    slices of a0, lanes 2,3 = those of a1 -- the slot layout of the (37,4) shape).  Inside a digit the two slices work as
    in the two-lane shapes of Gen (quotient digit broadcast and boundary column by DPP, modulus slice in VGPRs); between
    the digits the link of GenQ (lane 2 takes -m_i from lane 0 in row i; Cadj enters as the initial value of digit one's
    accumulators).  A squaring is 74 rows of 74 multiplies per lane -- half of the (37,4) kernel's 148 rows -- which is what
    counts when a batch is too small to fill the chip and the ladder's latency is the run time.
    A product is ONE pass too: the lanes of digit one keep a copy of a0 next to their own a1 (37 more registers, taken by
    DPP) and run two multiplier streams per row -- b0 against a1 and b1 against a0 -- into the same accumulators, so
    c1 = (a1 b0 + a0 b1 + Cadj - m) R^-1 comes out of one Montgomery reduction; the lanes of digit zero read zeros as
    their second stream.  74 rows of 111 multiplies instead of two passes of 74 rows of 74."""

    def __init__(self, WL=37):
        Gen.__init__(self, WL, 2)
        assert self.n_vreg and not self.flush
        self.H = 2 * WL                 # limbs of a digit
        self.WTslot = 4 * WL            # limbs of a slot
        self.WT = self.WTslot           # what the dispatcher's slot / constant addressing uses
        self.NPB = BLOCK // 4
        self.name = f"vm_asm_{WL}_64"
        self.nm4_tables = True     # STORET / MULVT / MULVT5: per-number window tables number-major
        self.sq_rows = True
        self.sq_rows_k = False
        # extra registers after the base map
        e = self.n_vgpr
        self.v_sh, self.v_l2mask, self.v_l3mask = e, e + 1, e + 2
        e += 3
        e = (e + 1) // 2 * 2
        self.v_d = e                    # pair (adjustment, 0)
        e += 2
        self.vR2 = e                    # a product's second multiplicand: the copy of a0 in the lanes of digit one
        e += WL
        self.v_ai2, self.v_ain2, self.v_arow2, self.v_bump2 = e, e + 1, e + 2, e + 3
        e += 4
        self.n_vgpr = e
        assert e <= 256, e
        self._xb = self.vX
        self.lanes_per_number = 4
        self.v_caddr = self.n_vgpr
        self.n_vgpr += 1
        self.alloc_row_regs()
        assert self.n_vgpr <= 256, self.n_vgpr
        # constants table in LDS: [WL][4 lanes] zero-extended limbs (0 | 0 | Cadj_j | Cadj_(WL+j)): the accumulators of a linked
        # pass START there (limb i of Cadj is the initial value of column i of digit one), as in GenQ3
        self.lds_c = self.lds_a + (self.WTslot + 1) * self.NPB * 4
        self.lds_z = self.lds_c + WL * 32       # 2 KB of zeros: the second multiplier stream of the lanes of digit zero
        self.lds_bytes = self.lds_z + 2048      # (a paired read reaches up to six rows past the pointer)
        assert self.lds_bytes < 65536
        # lane exchanges (GenQ8 below re-uses the rows with four slices per digit)
        self.dpp_link = "quad_perm:[0,1,0,3]"    # slice 0 of digit one <- slice 0 of digit zero
        self.dpp_bcast = "quad_perm:[0,0,2,2]"   # every slice of a digit <- its slice 0
        self.dpp_next = "quad_perm:[1,1,3,3]"    # slice s <- slice s + 1 of the same digit
        self.dpp_prev = "quad_perm:[0,0,2,2]"    # slice s <- slice s - 1
        self.dpp_copy0 = "quad_perm:[0,1,0,1]"   # the lanes of digit one <- the same slices of digit zero
        self.has_muls = True
        self.v_linkmask = self.v_l2mask          # -1 in the lane that takes the link
        self.v_d1mask = None                     # -1 in the lanes of digit one (GenQ4: l2mask | l3mask, formed where needed)

    def X(self, j):
        return f"v{self._xb + j}"

    def alloc_row_regs(self):
        """registers of the rows: the multiplier pairs of both streams (two rows are read at a time), the column that comes in
        from the slice above (a pair whose high word stays zero), the limb masks"""
        e = (self.n_vgpr + 1) // 2 * 2
        self.v_pa, self.v_pb, self.v_pa2, self.v_pb2, self.v_in = e, e + 2, e + 4, e + 6, e + 8
        e += 10
        self.v_mask28, self.v_rxmask, self.v_bump4 = e, e + 1, e + 2
        self.n_vgpr = e + 3

    def init_row_regs(self):
        g, e = self, self.e
        e(f"v_mov_b32 v{g.v_mask28}, {hex(MASK)}")
        e(f"v_and_b32 v{g.v_rxmask}, {hex(MASK)}, v{g.v_notlast}")     # a slice that has one above it takes that one's low column
        e(f"v_mov_b32 v{g.v_in + 1}, 0")
        e(f"v_mov_b32 v{g.v_bump4}, {4 * self.NPB * 4}")

    # slot / constant addressing uses the 4-lane slot width
    def slot_base(self):
        g, e = self, self.e
        e(f"s_mul_i32 s{g.s_t0}, s3, {self.WTslot}")
        e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, s{g.s_t0}")
        e(f"s_mul_i32 s{g.s_sbase}, s17, s{g.s_t0}")
        e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s10")
        e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s11")

    def const_base(self):
        g, e = self, self.e
        e(f"s_mul_i32 s{g.s_t0}, s17, {self.WTslot * 4}")
        e(f"s_add_u32 s{g.s_sbase}, s8, s{g.s_t0}")
        e(f"s_addc_u32 s{g.s_sbase + 1}, s9, 0")

    def prologue(self):
        g, e = self, self.e
        WL, NPB = self.WL, self.NPB
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 3, v0")                  # k4 = 2 d + s
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the constants table: k4 * 8
        e(f"v_lshrrev_b32 v{g.v_t2}, 2, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"v_mul_lo_u32 v{g.v_t4}, v{g.v_t1}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_t4}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_t4}, {WL * NPB * 4}, v{g.v_t1}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_t4}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t1}")
        e(f"v_and_b32 v{g.v_t4}, 1, v{g.v_t1}")           # s
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t4}")
        e(f"v_lshrrev_b32 v{g.v_sh}, 1, v{g.v_t1}")       # d (kept; a squaring uses it as the multiplier shift)
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t4}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_mov_b32 v{g.v_notlast}, v{g.v_isfirst}")     # two slices per digit: slice 0 is the first and the not-last one
        e(f"v_cmp_eq_u32 vcc, 2, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l2mask}, 0, -1, vcc")
        e(f"v_cmp_eq_u32 vcc, 3, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l3mask}, 0, -1, vcc")
        # modulus n (2 slices) -> LDS -> this lane's slice in VGPRs
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"v_cmp_gt_u32 vcc, {WL}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        for sgi in range(2):
            e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
        # constants table: thread t < WL writes row t = (0 | 0 | Cadj_t | Cadj_(WL+t)), zero-extended to 64 bits
        e(f"global_load_dword v{g.v_p0}, v{g.v_t3}, s[6:7] offset:{4 * self.H}")
        e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{4 * self.H + 4 * WL}")
        e(f"v_mov_b32 v{g.v_p0 + 1}, 0")
        e(f"v_mov_b32 v{g.v_p1 + 1}, 0")
        e(f"v_mov_b32 v{g.v_y0}, 0")
        e(f"v_mov_b32 v{g.v_y0 + 1}, 0")
        e(f"v_lshlrev_b32 v{g.v_t4}, 3, v{g.v_t3}")        # t * 32
        e("s_waitcnt vmcnt(0)")
        e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c}")
        e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8}")
        e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p0)} offset:{self.lds_c + 16}")
        e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p1)} offset:{self.lds_c + 24}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e(f"v_mov_b32 v{g.v_p0}, 0")
        e(f"ds_write_b32 v{g.v_t3}, v{g.v_p0} offset:{self.lds_z}")      # every thread zeroes two words of the zero rows
        e(f"ds_write_b32 v{g.v_t3}, v{g.v_p0} offset:{self.lds_z + 1024}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(WL):
            e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")
        self.init_row_regs()

    def read_pair(self, dst, ptr, first_row):
        """multipliers of rows first_row, first_row + 1 past the row pointer -> the register pair dst"""
        rowb = self.NPB * 4
        if rowb == 256:
            self.e(f"ds_read2st64_b32 v[{dst}:{dst + 1}], v{ptr} offset0:{first_row} offset1:{first_row + 1}")
        else:
            assert rowb % 4 == 0 and (first_row + 1) * rowb // 4 <= 255
            self.e(f"ds_read2_b32 v[{dst}:{dst + 1}], v{ptr} offset0:{first_row * rowb // 4} offset1:{(first_row + 1) * rowb // 4}")

    def chain_steps(self, link):
        """the dependent instructions between column 0 of a row and the quotient digit in every slice"""
        g = self
        m = f"v{g.v_m}"
        chain = [f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        if link:
            chain += [f"v_and_b32_dpp v{g.v_d}, {m}, v{g.v_mask28} {self.dpp_link} row_mask:0xf bank_mask:0xf",
                      f"v_mad_i64_i32 {self.T(0)}, vcc, v{g.v_d}, v{g.v_linkmask}, {self.T(0)}",    # digit one, slice 0: T0 -= m of digit zero
                      f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        chain.append(f"v_and_b32_dpp {m}, {m}, v{g.v_mask28} {self.dpp_bcast} row_mask:0xf bank_mask:0xf")
        return chain

    def row(self, a, link, a2=None):
        """one Montgomery row modulo n in the lanes of a number.  a: this row's multiplier; link: the link lane takes -m_i of digit
        zero into column 0 (its Cadj limb has been in the accumulator since the pass began); a2: the second multiplier stream of a
        product (against the multiplicand copy vR2).
        Column 0 is complete after the FIRST multiplies of pass A, so the quotient digit, the link hop and the broadcast to the
        digit's upper slices -- a chain of dependent instructions -- start right away and are spread between the remaining
        multiplies of pass A: no s_nop for the DPP hazard, nothing waits on a result in flight.  At one wave per SIMD every
        instruction of the wave, scalar or not, takes an issue slot, so the row is kept to what cannot be avoided:
          - a limb mask and the lane exchange that follows it are ONE instruction (v_and_b32 with a DPP source);
          - every slice keeps the carry of its own column 0 (the high part of y0) and hands only the LOW 28 bits to the slice
            below, where they are the new top column: the same value as moving the whole column, without masking the carry by
            lane, and the incoming column is a 32-bit register whose pair partner is a constant zero;
          - multipliers are read two rows at a time."""
        g, e = self, self.e
        WL = self.WL
        N = lambda j: f"v{g.v_N + j}"
        m = f"v{g.v_m}"
        IN = self.P(g.v_in)
        chain = self.chain_steps(link)
        muls = []
        for j in range(WL):
            muls.append((self.T(j), f"v{a}", self.X(j), IN if j == WL - 1 else self.T(j)))
            if a2 is not None:
                muls.append((self.T(j), f"v{a2}", f"v{g.vR2 + j}", self.T(j)))
        head = 1 if a2 is None else 2                    # multiplies that complete column 0
        gap = min(4, (len(muls) - head) // len(chain))
        # (10-limb slices: a linked squaring row has 9 multiplies for a chain of 5 -- one multiply between two steps and a wait state
        # in front of every step that reads its operand through DPP)
        assert gap >= 1
        self.align8()
        k = 0
        for _ in range(head):
            self.mad(*muls[k])
            k += 1
        for step in chain:
            for _ in range(gap):
                self.mad(*muls[k])
                k += 1
            if gap < 2 and "_dpp" in step:
                e("s_nop 0")
            e(step)
        while k < len(muls):
            self.mad(*muls[k])
            k += 1
        self.mad(self.P(g.v_y0), m, N(0), self.T(0))
        self.mad(self.T(0), m, N(1), self.T(1))
        e(f"v_lshrrev_b64 {self.P(g.v_c)}, {LB}, {self.P(g.v_y0)}")
        for j in range(2, WL):
            self.mad(self.T(j - 1), m, N(j), self.T(j))
            if j == 4:
                e(f"v_lshl_add_u64 {self.T(0)}, {self.T(0)}, 0, {self.P(g.v_c)}")
        e(f"v_and_b32_dpp v{g.v_in}, v{g.v_y0}, v{g.v_rxmask} {self.dpp_next} row_mask:0xf bank_mask:0xf{getattr(self, 'dpp_next_tail', '')}")

    def row_pair(self, cur, nxt, link, ahead, bump, cur2=None, nxt2=None, count=False):
        """two rows on the multiplier pair(s) cur; the pair(s) of the two rows `ahead` rows past the pointer are fetched into nxt;
        bump: rows the pointers move on afterwards (2 before the loop, 4 inside it); count: the loop counter moves here (the 4-byte
        scalar instructions are paired so that the multiply streams stay on 8-byte boundaries without padding)"""
        g, e = self, self.e
        rowb = self.NPB * 4
        e("s_waitcnt lgkmcnt(0)")
        if count:
            e("s_add_u32 s19, s19, 1")
        self.read_pair(nxt, g.v_arow, ahead)
        if cur2 is not None:
            self.read_pair(nxt2, g.v_arow2, ahead)
        if bump == 4:
            e(f"v_add_u32 v{g.v_arow}, v{g.v_bump4}, v{g.v_arow}")
            if cur2 is not None:
                e(f"v_lshl_add_u32 v{g.v_arow2}, v{g.v_bump2}, 1, v{g.v_arow2}")
        elif bump:
            assert bump == 2 and not count
            e(f"v_add_u32 v{g.v_arow}, {2 * rowb}, v{g.v_arow}")
            if cur2 is not None:
                e(f"v_add_u32 v{g.v_arow2}, v{g.v_bump2}, v{g.v_arow2}")
            e("s_nop 0")
        self.row(cur, link, cur2)
        self.row(cur + 1, link, None if cur2 is None else cur2 + 1)

    def passes(self, tag, two_streams):
        """H linked rows: T <- (b0 stream) * X [+ (b1 stream | zeros) * vR2] * R^-1, accumulators starting at the constants"""
        g, e = self, self.e
        WL, H = self.WL, self.H
        row = self.NPB * 4
        assert H % 2 == 0
        for j in range(WL):                                      # accumulators <- (0 | 0 | Cadj_j | Cadj_(WL+j)) by lane
            e(f"ds_read_b64 {self.T(j)}, v{g.v_caddr} offset:{self.lds_c + self.lanes_per_number * 8 * j}")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        self.read_pair(g.v_pa, g.v_arow, 0)
        if two_streams:
            # stream two: rows H.. of the a column (b1) in the lanes of digit one, the zero rows in the lanes of digit zero
            e(f"v_add_u32 v{g.v_arow2}, {H * row}, v{g.v_aread}")
            e(f"v_mov_b32 v{g.v_t2}, {self.lds_z}")
            e(f"v_xor_b32 v{g.v_t2}, v{g.v_t2}, v{g.v_arow2}")
            if g.v_d1mask is None:
                e(f"v_or_b32 v{g.v_t3}, v{g.v_l2mask}, v{g.v_l3mask}")
            else:
                e(f"v_mov_b32 v{g.v_t3}, v{g.v_d1mask}")
            e(f"v_and_b32 v{g.v_t2}, v{g.v_t2}, v{g.v_t3}")
            e(f"v_xor_b32 v{g.v_arow2}, {self.lds_z}, v{g.v_t2}")      # digit one: aread + H rows; digit zero: lds_z
            e(f"v_and_b32 v{g.v_bump2}, {2 * row}, v{g.v_t3}")
            self.read_pair(g.v_pa2, g.v_arow2, 0)
        self.row_loop(tag, True, two_streams)

    def row_loop(self, tag, link, two_streams):
        """the H rows of a pass; the first multiplier pair(s) are on their way"""
        g, e = self, self.e
        WL, H = self.WL, self.H
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_mov_b32 v{g.v_in}, {self.Tlo(WL - 1)}")            # the top column's constant comes in as its first addend
        s2 = lambda c2, n2: dict(cur2=c2, nxt2=n2) if two_streams else {}
        pairs = H // 2
        A, B = (g.v_pa, getattr(g, 'v_pa2', None)), (g.v_pb, getattr(g, 'v_pb2', None))
        if pairs % 2:
            self.row_pair(A[0], B[0], link, 2, 2, **s2(A[1], B[1]))
            A, B = B, A
        e("s_mov_b32 s19, 0")
        e(".p2align 6")
        e(f"L_q{tag}:")
        self.row_pair(A[0], B[0], link, 2, 0, count=True, **s2(A[1], B[1]))
        self.row_pair(B[0], A[0], link, 4, 4, **s2(B[1], A[1]))
        e(f"s_cmp_lt_u32 s19, {pairs // 2}")
        e(f"s_cbranch_scc1 L_q{tag}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_mov_b32 {self.Tlo(WL - 1)}, v{g.v_in}")            # the last row's incoming column is the top column of the result
        e(f"v_mov_b32 {self.Thi(WL - 1)}, 0")

    def normalize(self, dst=None):
        """accumulators -> limbs of the new x (dst(j): where limb j goes; default the multiplicand registers).  Two lanes per digit: both run the plain sequential carry of a one-lane number
        over their own 37 columns (3 instructions a limb: every lane owns whole columns), then the lower slice's carry out of its
        top column -- up to 2^36 -- enters the upper slice's two lowest limbs (limb 1 stays lazy by < 2^9: multiplicands and
        multipliers have that headroom).  125 instructions where the carry-save form of the generic multi-lane shapes takes
        350; the upper slice's top limb keeps the digit's excess, as everywhere."""
        g, e = self, self.e
        WL = self.WL
        M = hex(MASK)
        c = self.P(g.v_c)
        e(f"v_not_b32 v{g.v_t2}, v{g.v_notlast}")                       # -1 in the top slice of a digit
        e(f"v_or_b32 v{g.v_t2}, {M}, v{g.v_t2}")                         # top-limb mask: 28 bits below the top slice, everything in it
        e(f"v_not_b32 v{g.v_t1}, v{g.v_isfirst}")                       # -1 in every slice that has one below it
        dst = dst or self.X
        # (shift before mask: dst(j) may be the low half of T(j) itself)
        e(f"v_lshrrev_b64 {c}, {LB}, {self.T(0)}")
        e(f"v_and_b32 {dst(0)}, {M}, {self.Tlo(0)}")
        for j in range(1, WL):
            e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")
            if j < WL - 1:
                e(f"v_and_b32 {dst(j)}, {M}, {self.Tlo(j)}")
            else:
                e(f"v_and_b32 {dst(j)}, {self.Tlo(j)}, v{g.v_t2}")
        # carry out of a slice -> the next slice of the same digit (it lands in that slice's two lowest limbs and goes no
        # further: the slices' own carries out were taken before it came in)
        e("s_nop 1")
        e(f"v_mov_b32_dpp v{g.v_p0}, v{g.v_c} {self.dpp_prev} row_mask:0xf bank_mask:0xf")
        e(f"v_mov_b32_dpp v{g.v_p0 + 1}, v{g.v_c + 1} {self.dpp_prev} row_mask:0xf bank_mask:0xf")
        e(f"v_and_b32 v{g.v_p0}, v{g.v_p0}, v{g.v_t1}")
        e(f"v_and_b32 v{g.v_p0 + 1}, v{g.v_p0 + 1}, v{g.v_t1}")
        e(f"v_mov_b32 v{g.v_c}, {dst(0)}")
        e(f"v_mov_b32 v{g.v_c + 1}, 0")
        e(f"v_lshl_add_u64 {c}, {c}, 0, {self.P(g.v_p0)}")               # limb 0 + carry in (zero in the lower slice)
        e(f"v_and_b32 {dst(0)}, {M}, v{g.v_c}")
        e(f"v_lshrrev_b64 {c}, {LB}, {c}")
        e(f"v_add_u32 {dst(1)}, {dst(1)}, v{g.v_c}")

    def montsq(self):
        g, e = self, self.e
        e("L_montsq:")
        # digit one computes 2 a0 a1: its lanes double their MULTIPLICAND once (a 29-bit limb times a 28-bit multiplier over
        # 74 rows, plus the reduction, stays below 2^64 per column) instead of the multiplier in every row
        for j in range(self.WL):
            e(f"v_lshlrev_b32 {self.X(j)}, v{g.v_sh}, {self.X(j)}")
        self.passes("s", False)
        self.normalize()
        e("s_branch L_next")

    def montmul(self, to_slot=False):
        g, e = self, self.e
        e("L_montmuls:" if to_slot else "L_montmul:")
        e("s_nop 1")
        for j in range(self.WL):                                 # digit one <- the slices of a0 (digit zero: whatever, times zero)
            e(f"v_mov_b32_dpp v{g.vR2 + j}, {self.X(j)} {self.dpp_copy0} row_mask:0xf bank_mask:0xf")
        # digit zero: t = a0 b0 R^-1 (second stream: zeros); digit one: c1 = (a1 b0 + a0 b1 + Cadj - m) R^-1
        self.passes("ms" if to_slot else "m", True)
        if not to_slot:
            self.normalize()
            e("s_branch L_next")
            return
        # VM_MULS: the limbs of the product go to the registers of the multiplicand copy (dead after the rows) and from there to
        # the slot the operand came from; x stays as it is
        self._xb = self.vR2
        self.normalize()
        self._xb = self.vX
        e(f"v_mov_b32 v{g.v_addr}, v{g.v_goff}")
        for j in range(self.WL):
            e(f"global_store_dword v{g.v_addr}, v{g.vR2 + j}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            if j != self.WL - 1:
                e(f"v_add_u32 v{g.v_addr}, s3, v{g.v_addr}")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montmul(to_slot=True)
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenS4(GenQ4):
    """ONE digit sliced over the four lanes of a quad: the 37-limb primes of a 2048-bit key as 40-limb moduli, four lanes of 10 limbs per
    number -- the plain Montgomery contract of Gen(10, 4) (same slots, constants, opcodes, radix 2^(28 * 40)) on the rows of the
    four-slice pair kernel GenQ8 without the second digit: the quotient digit of slice 0 and its broadcast interleaved with the first
    multiplies of a row, a limb mask and the lane exchange that follows it in one instruction, multipliers read two rows at a time, the
    carries of a slice taken sequentially in its own lane.  27 instructions a row (20 of them multiplies) where the generic four-lane
    rows of Gen have 36: for ladders modulo the primes whose LATENCY is the run time (plan::prime_lanes)."""

    def __init__(self, WL=10):
        Gen.__init__(self, WL, 4)
        assert self.n_vreg and not self.flush
        self.H = 4 * WL
        self.WTslot = self.WT
        self.name = f"vm_asm_{WL}_4"
        self.nm4_tables = True
        self.sq_rows = True              # (the dispatcher's L_sqr stages x and branches to L_montsq)
        self.sq_rows_k = False
        self.lanes_per_number = 4
        self.has_muls = False
        self.alloc_row_regs()
        assert self.n_vgpr <= 256, self.n_vgpr
        # a paired read reaches up to six rows past the pointer: rows of padding behind the multiplier columns
        self.lds_bytes = self.lds_a + (self.WT + 8) * self.NPB * 4
        assert self.lds_bytes < 65536
        self.dpp_bcast = "quad_perm:[0,0,0,0]"
        self.dpp_next = "quad_perm:[1,2,3,3]"
        self.dpp_prev = "quad_perm:[0,0,1,2]"
        self._xb = self.vX

    def prologue(self):
        Gen.prologue(self)
        self.init_row_regs()

    def passes(self, tag, two_streams=False):
        g, e = self, self.e
        assert not two_streams
        for j in range(self.WL):
            e(f"v_mov_b64 {self.T(j)}, 0")
        e(f"v_mov_b32 v{g.v_arow}, v{g.v_aread}")
        self.read_pair(g.v_pa, g.v_arow, 0)
        self.row_loop(tag, False, False)

    def montsq(self):
        # x <- x x R^-1: the multipliers are x's own limbs, staged by the dispatcher (no symmetry taken: every lane would idle for
        # the half it skips)
        self.e("L_montsq:")
        self.passes("s")
        self.normalize()
        self.e("s_branch L_next")

    def montmul(self, to_slot=False):
        assert not to_slot
        self.e("L_montmul:")
        self.passes("m")
        self.normalize()
        self.e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenQ8(GenQ4):
    """The pair kernel with every digit sliced over FOUR lanes: 8 lanes per number (lanes 0..3 = the 19-limb slices of a0, lanes
    4..7 = those of a1; a digit is 76 limbs -- the 74 limbs of a 2048-bit n padded: the host converts between the two radices
    R_74 and R_76 with one product on the way in and one on the way out).  The rows, the one-pass product and the carry
    normalisation are GenQ4's with the exchanges of a four-slice number (one quad per digit; the link and the copy of a0 cross to
    the quad above by row_shr:4).  A squaring is 76 rows of 38 multiplies per lane where GenQ4 has 74 rows of 74: what counts
    when a batch is far too small to fill the chip and the LATENCY of one ladder is the run time (a rank's 6 144 units of the
    sharded threshold flow at N = 8)."""

    def __init__(self, WL=19):
        Gen.__init__(self, WL, 4)
        assert self.n_vreg and not self.flush
        self.H = 4 * WL
        self.WTslot = 8 * WL
        self.WT = self.WTslot
        self.NPB = BLOCK // 8
        self.name = f"vm_asm_{WL}_96"
        self.sq_rows = True
        self.sq_rows_k = False
        self.lanes_per_number = 8
        self.lds_a = (4 * self.WLp * 4 + 15) // 16 * 16
        e = self.n_vgpr
        self.v_sh, self.v_l2mask, self.v_l3mask = e, e + 1, e + 2      # l2mask: the link lane (k8 == 4); l3mask: digit one (k8 >= 4)
        e += 3
        e = (e + 1) // 2 * 2
        self.v_d = e
        e += 2
        self.vR2 = e
        e += WL
        self.v_ai2, self.v_ain2, self.v_arow2, self.v_bump2 = e, e + 1, e + 2, e + 3
        e += 4
        self._xb = self.vX
        self.v_caddr = e
        e += 1
        self.n_vgpr = e
        self.alloc_row_regs()
        assert self.n_vgpr <= 256, self.n_vgpr
        self.lds_c = self.lds_a + (self.WTslot + 1) * self.NPB * 4
        self.lds_z = self.lds_c + WL * 64
        self.lds_bytes = self.lds_z + 1024
        assert self.lds_bytes < 65536
        self.has_muls = True
        self.dpp_link = "row_shr:4"
        self.dpp_bcast = "quad_perm:[0,0,0,0]"
        self.dpp_next = "quad_perm:[1,2,3,3]"
        self.dpp_prev = "quad_perm:[0,0,1,2]"
        self.dpp_copy0 = "row_shr:4"
        self.v_linkmask = self.v_l2mask
        self.v_d1mask = self.v_l3mask

    def prologue(self):
        g, e = self, self.e
        WL, NPB = self.WL, self.NPB
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 7, v0")                  # k8 = 4 d + s
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the constants table: k8 * 8
        e(f"v_lshrrev_b32 v{g.v_t2}, 3, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"v_mul_lo_u32 v{g.v_t4}, v{g.v_t1}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_t4}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_t4}, {WL * NPB * 4}, v{g.v_t1}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_t4}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t1}")
        e(f"v_and_b32 v{g.v_t4}, 3, v{g.v_t1}")           # s
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t4}")
        e(f"v_lshrrev_b32 v{g.v_sh}, 2, v{g.v_t1}")       # d (kept; a squaring uses it as the multiplicand shift)
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t4}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_cmp_ne_u32 vcc, 3, v{g.v_t4}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_notlast}, 0, -1, vcc")
        e(f"v_cmp_eq_u32 vcc, 4, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l2mask}, 0, -1, vcc")
        e(f"v_cmp_le_u32 vcc, 4, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l3mask}, 0, -1, vcc")
        # modulus n (4 slices) -> LDS -> this lane's slice in VGPRs
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"v_cmp_gt_u32 vcc, {WL}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        for sgi in range(4):
            e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
        # constants table: thread t < WL writes row t = (0 0 0 0 | Cadj_t, Cadj_(WL+t), Cadj_(2WL+t), Cadj_(3WL+t)), zero-extended
        e(f"v_lshlrev_b32 v{g.v_t4}, 4, v{g.v_t3}")        # t * 64
        e(f"v_mov_b32 v{g.v_y0}, 0")
        e(f"v_mov_b32 v{g.v_y0 + 1}, 0")
        e(f"v_mov_b32 v{g.v_p0 + 1}, 0")
        for sgi in range(4):
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * sgi}")
        for sgi in range(4):
            e(f"global_load_dword v{g.v_p0}, v{g.v_t3}, s[6:7] offset:{4 * self.H + 4 * WL * sgi}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p0)} offset:{self.lds_c + 32 + 8 * sgi}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e(f"v_mov_b32 v{g.v_p0}, 0")
        e(f"ds_write_b32 v{g.v_t3}, v{g.v_p0} offset:{self.lds_z}")      # every thread zeroes one word of the zero rows
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(WL):
            e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")
        self.init_row_regs()


class GenQ16(GenQ8):
    """The pair kernel with every digit sliced over EIGHT lanes: 16 lanes per number -- one DPP row (lanes 0..7 = the 10-limb slices of
    a0, lanes 8..15 = those of a1; a digit is 80 limbs: the 74 limbs of a 2048-bit n padded, radix R_80, the host changes radix with
    one product on the way in and one on the way out as for GenQ8).  Rows, one-pass product and carries are GenQ4's with the exchanges
    of a row: neighbours by row_shl:1 / row_shr:1 (what crosses a digit is masked, what comes from outside the row reads as zero), the
    link and the copy of a0 by row_shr:8, the quotient digit's broadcast in two steps (its quad, then the quad above takes it from
    four lanes down).  A squaring is 80 rows of 20 multiplies per lane where GenQ8 has 76 rows of 38: for shared-exponent ladders
    modulo n^2 of at most 2 048 numbers (the verifier's E^n and y^n, small Encrypt batches), whose latency is the run time -- 6.4 us a
    squaring against 7.4: the quotient chain of a row (six dependent steps, one multiply apart) is not hidden any more."""

    def __init__(self, WL=10):
        Gen.__init__(self, WL, 8)
        assert self.n_vreg and not self.flush
        self.H = 8 * WL
        self.WTslot = 16 * WL
        self.WT = self.WTslot
        self.NPB = BLOCK // 16
        self.name = f"vm_asm_{WL}_128"
        self.sq_rows = True
        self.sq_rows_k = False
        self.lanes_per_number = 16
        self.lds_a = (8 * self.WLp * 4 + 15) // 16 * 16
        e = self.n_vgpr
        self.v_sh, self.v_l2mask, self.v_l3mask = e, e + 1, e + 2      # l2mask: the link lane (k16 == 8); l3mask: digit one (k16 >= 8)
        e += 3
        e = (e + 1) // 2 * 2
        self.v_d = e
        e += 2
        self.vR2 = e
        e += WL
        self.v_ai2, self.v_ain2, self.v_arow2, self.v_bump2 = e, e + 1, e + 2, e + 3
        e += 4
        self._xb = self.vX
        self.v_caddr = e
        e += 1
        self.n_vgpr = e
        self.alloc_row_regs()
        assert self.n_vgpr <= 256, self.n_vgpr
        self.lds_c = self.lds_a + (self.WTslot + 1) * self.NPB * 4
        self.lds_z = self.lds_c + WL * 128
        self.lds_bytes = self.lds_z + 1024
        assert self.lds_bytes < 65536
        self.has_muls = True
        self.dpp_link = "row_shr:8"
        self.dpp_bcast = None                     # two steps: chain_steps
        self.dpp_next = "row_shl:1"
        self.dpp_next_tail = " bound_ctrl:1"      # lane 15 reads past the row: zero, not its stale register
        self.dpp_prev = "row_shr:1"
        self.dpp_copy0 = "row_shr:8"
        self.v_linkmask = self.v_l2mask
        self.v_d1mask = self.v_l3mask

    def chain_steps(self, link):
        g = self
        m = f"v{g.v_m}"
        chain = [f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        if link:
            chain += [f"v_and_b32_dpp v{g.v_d}, {m}, v{g.v_mask28} {self.dpp_link} row_mask:0xf bank_mask:0xf",
                      f"v_mad_i64_i32 {self.T(0)}, vcc, v{g.v_d}, v{g.v_linkmask}, {self.T(0)}",
                      f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        # slice 0 of the digit -> its quad, then the digit's upper quad (banks 1 and 3 of the row) takes it from four lanes down
        chain.append(f"v_and_b32_dpp {m}, {m}, v{g.v_mask28} quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf")
        chain.append(f"v_mov_b32_dpp {m}, {m} row_shr:4 row_mask:0xf bank_mask:0xa")
        return chain

    def prologue(self):
        g, e = self, self.e
        WL, NPB = self.WL, self.NPB
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 15, v0")                 # k16 = 8 d + s
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the constants table: k16 * 8
        e(f"v_lshrrev_b32 v{g.v_t2}, 4, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"v_mul_lo_u32 v{g.v_t4}, v{g.v_t1}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_t4}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_t4}, {WL * NPB * 4}, v{g.v_t1}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_t4}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t1}")
        e(f"v_and_b32 v{g.v_t4}, 7, v{g.v_t1}")           # s
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t4}")
        e(f"v_lshrrev_b32 v{g.v_sh}, 3, v{g.v_t1}")       # d (kept; a squaring uses it as the multiplicand shift)
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t4}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_cmp_ne_u32 vcc, 7, v{g.v_t4}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_notlast}, 0, -1, vcc")
        e(f"v_cmp_eq_u32 vcc, 8, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l2mask}, 0, -1, vcc")
        e(f"v_cmp_le_u32 vcc, 8, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_l3mask}, 0, -1, vcc")
        # modulus n (8 slices) -> LDS -> this lane's slice in VGPRs
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"v_cmp_gt_u32 vcc, {WL}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        for sgi in range(8):
            e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
        # constants table: thread t < WL writes row t = (eight zeros | Cadj_t, Cadj_(WL+t), ..., Cadj_(7 WL+t)), zero-extended to 64 bits
        e(f"v_lshlrev_b32 v{g.v_t4}, 5, v{g.v_t3}")        # t * 128
        e(f"v_mov_b32 v{g.v_y0}, 0")
        e(f"v_mov_b32 v{g.v_y0 + 1}, 0")
        e(f"v_mov_b32 v{g.v_p0 + 1}, 0")
        for sgi in range(8):
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * sgi}")
        for sgi in range(8):
            e(f"global_load_dword v{g.v_p0}, v{g.v_t3}, s[6:7] offset:{4 * self.H + 4 * WL * sgi}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p0)} offset:{self.lds_c + 64 + 8 * sgi}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e(f"v_mov_b32 v{g.v_p0}, 0")
        e(f"ds_write_b32 v{g.v_t3}, v{g.v_p0} offset:{self.lds_z}")      # every thread zeroes one word of the zero rows
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(WL):
            e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")
        self.init_row_regs()


class GenQ6(GenQ4):
    """The three-digit kernel with every digit sliced over TWO lanes: 8 lanes per number for moduli N = n^3 whose digit does not fit
    one lane (H = 110: 3072-bit keys) or whose batch is bound by one ladder's latency (H = 74).  Lane k8 of a number:
        0, 1 = the slices of a0 | 2, 3 = those of a2 | 4, 5 = those of a1 | 6, 7 = the helper digit
    (a1 next to the helper: the helper takes its copy of a1 by a quad_perm confined to that quad with bank_mask; the quotient links
    a0 -> a1 -> a2 cross the quads by row_shr:4 and row_shl:2).  Rows are GenQ4's (modulus slice in VGPRs, quotient digit
    broadcast and boundary column inside the lane pair of a digit) with GenQ3's two hops in the chain; squaring one pass, product
    two passes as GenQ3 at H = 74; carries inside a digit as GenQ4.normalize.  Shared-exponent programs only (no number-major
    tables)."""

    def __init__(self, WL):
        Gen.__init__(self, WL, 2)
        assert self.n_vreg and not self.flush
        self.H = 2 * WL
        self.WTslot = 6 * WL
        self.WT = self.WTslot
        self.NPB = BLOCK // 8
        self.name = f"vm_asm_{WL}_112"
        self.sq_rows = True
        self.sq_rows_k = False
        self.has_muls = False
        self.lanes_per_number = 8
        self.lds_a = (2 * self.WLp * 4 + 15) // 16 * 16
        e = self.n_vgpr
        for nm in ["sh", "l1mask", "l2mask", "hmask", "t5", "caddr"]:
            setattr(self, "v_" + nm, e)
            e += 1
        e = (e + 1) // 2 * 2
        self.v_d = e
        e += 2
        self.v_nbase = self.v_d         # (the modulus slice's LDS address is needed in the prologue only, v_d in the rows only)
        self._xb = self.vX
        self.n_vgpr = e
        assert e <= 256, e
        # the registers of GenQ4's rows come from slots of the base map that this kernel does not use (the two-lane squaring's
        # scalars, the single multiplier registers and the alignment gap after them)
        pool = sorted([self.v_ai, self.v_ain, self.v_k, 3 * WL + 5, self.v_mk, self.v_mult, self.v_km2, self.v_islast] +
                      ([self.v_y0 - 1] if (3 * WL + 10) % 2 else []))
        pairs = []
        for r in list(pool):
            if r % 2 == 0 and r in pool and r + 1 in pool and len(pairs) < 3:
                pairs.append(r)
                pool.remove(r)
                pool.remove(r + 1)
        assert len(pairs) == 3 and len(pool) >= 3, (pairs, pool)
        self.v_pa, self.v_pb, self.v_in = pairs
        self.v_mask28, self.v_rxmask, self.v_bump4 = pool[:3]
        del self.v_ai, self.v_ain
        self.lds_c = self.lds_a + (self.WTslot + 1) * self.NPB * 4
        self.lds_bytes = self.lds_c + WL * 64
        assert self.lds_bytes < 65536
        self.dpp_bcast = "quad_perm:[0,0,2,2]"
        self.dpp_next = "quad_perm:[1,1,3,3]"
        self.dpp_prev = "quad_perm:[0,0,2,2]"
        self.npad = (self.H + 1) // 2 * 2

    def set_exec8(self, mask8):
        w = sum(mask8 << (8 * i) for i in range(4))
        self.e(f"s_mov_b32 exec_lo, {hex(w)}")
        self.e(f"s_mov_b32 exec_hi, {hex(w)}")

    def mask_digit_lanes(self, on):
        if on:
            self.set_exec8(0x3f)
        else:
            self.e("s_mov_b64 exec, -1")

    def prologue(self):
        g, e = self, self.e
        WL, NPB, H = self.WL, self.NPB, self.H
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 7, v0")                  # k8
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the constants table
        e(f"v_lshrrev_b32 v{g.v_t2}, 3, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        # slot position of the lane: digit dd = (0, 2, 1, 2)[k8 >> 1] (the helper aliases digit two; its stores are masked), slice s
        e(f"v_lshrrev_b32 v{g.v_t4}, 1, v{g.v_t1}")       # q = k8 >> 1
        e(f"v_and_b32 v{g.v_t5}, 1, v{g.v_t4}")           # t = q & 1
        e(f"v_xor_b32 v{g.v_p0}, 1, v{g.v_t5}")           # t ^ 1
        e(f"v_lshrrev_b32 v{g.v_p0 + 1}, 1, v{g.v_t4}")   # q >> 1
        e(f"v_and_b32 v{g.v_p0}, v{g.v_p0}, v{g.v_p0 + 1}")
        e(f"v_lshl_or_b32 v{g.v_t4}, v{g.v_t5}, 1, v{g.v_p0}")   # dd
        e(f"v_and_b32 v{g.v_t5}, 1, v{g.v_t1}")           # s
        e(f"v_lshl_or_b32 v{g.v_t4}, v{g.v_t4}, 1, v{g.v_t5}")   # kk = 2 dd + s
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"v_mul_lo_u32 v{g.v_p0}, v{g.v_t4}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_p0}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_p0}, {WL * NPB * 4}, v{g.v_t4}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_p0}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t4}")
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t5}")
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t5}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_mov_b32 v{g.v_notlast}, v{g.v_isfirst}")
        for val, reg in ((4, g.v_l1mask), (2, g.v_l2mask)):
            e(f"v_cmp_eq_u32 vcc, {val}, v{g.v_t1}")
            e("s_nop 1")
            e(f"v_cndmask_b32 v{reg}, 0, -1, vcc")
        e(f"v_cmp_le_u32 vcc, 6, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_hmask}, 0, -1, vcc")
        e(f"v_add_u32 v{g.v_sh}, 2, v{g.v_t1}")
        e(f"v_bfe_u32 v{g.v_sh}, v{g.v_sh}, 2, 1")         # 1 in the lanes of a1 and a2 (k8 = 2 .. 5): they double their multiplicand
        # modulus n (2 slices) -> LDS -> this lane's slice in VGPRs
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"v_cmp_gt_u32 vcc, {WL}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        for sgi in range(2):
            e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
        # constants table: thread t < WL writes row t = (0 0 | C2_t C2_(WL+t) | C1_t C1_(WL+t) | 0 0), zero-extended
        e(f"v_lshlrev_b32 v{g.v_t4}, 4, v{g.v_t3}")        # t * 64
        e(f"v_lshlrev_b32 v{g.v_t5}, 1, v{g.v_t3}")        # t * 8: the pair (C1_t, C2_t)
        e(f"v_mov_b32 v{g.v_y0}, 0")
        e(f"v_mov_b32 v{g.v_y0 + 1}, 0")
        for k in (0, 1, 6, 7):
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * k}")
        for sgi in range(2):
            e(f"global_load_dwordx2 {self.P(g.v_p0)}, v{g.v_t5}, s[6:7] offset:{4 * self.npad + 8 * WL * sgi}")
            e("s_waitcnt vmcnt(0)")
            e(f"v_mov_b32 v{g.v_y0}, v{g.v_p0 + 1}")       # (C2, 0)
            e(f"v_mov_b32 v{g.v_p0 + 1}, 0")               # (C1, 0)
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p0)} offset:{self.lds_c + 8 * (4 + sgi)}")
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * (2 + sgi)}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e("s_barrier")
        for j in range(WL):
            e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")
        self.init_row_regs()

    def chain_steps(self, link2):
        """Link one: slice 0 of a1 takes -m of a0; link two (link2): slice 0 of a2 takes -m of a1; every quotient digit is then
        broadcast to the digit's upper slice."""
        g = self
        m = f"v{g.v_m}"
        chain = [f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        for ctrl, mask in [("row_shr:4", g.v_l1mask)] + ([("row_shl:2", g.v_l2mask)] if link2 else []):
            chain += [f"v_and_b32_dpp v{g.v_d}, {m}, v{g.v_mask28} {ctrl} row_mask:0xf bank_mask:0xf",
                      f"v_mad_i64_i32 {self.T(0)}, vcc, v{g.v_d}, v{mask}, {self.T(0)}",
                      f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        chain.append(f"v_and_b32_dpp {m}, {m}, v{g.v_mask28} {self.dpp_bcast} row_mask:0xf bank_mask:0xf")
        return chain

    def passes(self, tag, off_d, off_h, link2):
        """H rows (GenQ4's, with this kernel's chain).  off_d / off_h: first row of the multiplier stream read by the digit lanes /
        by the helper lanes."""
        g, e = self, self.e
        WL, H = self.WL, self.H
        row = self.NPB * 4
        for j in range(WL):                                      # accumulators <- (0 | C2 | C1 | 0) by lane
            e(f"ds_read_b64 {self.T(j)}, v{g.v_caddr} offset:{self.lds_c + 64 * j}")
        e(f"v_and_b32 v{g.v_t1}, {(off_h - off_d) * row}, v{g.v_hmask}")
        e(f"v_add_u32 v{g.v_arow}, v{g.v_aread}, v{g.v_t1}")
        if off_d:
            e(f"v_add_u32 v{g.v_arow}, {off_d * row}, v{g.v_arow}")
        self.read_pair(g.v_pa, g.v_arow, 0)
        self.row_loop(tag, link2, False)

    def carry_X2(self):
        """lazy limbs (sums of up to three canonical limbs) of the active lanes -> canonical again inside every slice; the lower
        slice's carry goes into limb 0 of the upper one (lazy by a few units); the upper slice's top limb keeps the excess"""
        g, e = self, self.e
        WL = self.WL
        M = hex(MASK)
        e(f"v_not_b32 v{g.v_t2}, v{g.v_notlast}")
        e(f"v_or_b32 v{g.v_t2}, {M}, v{g.v_t2}")                         # top-limb mask
        e(f"v_not_b32 v{g.v_t3}, v{g.v_isfirst}")
        for j in range(WL):
            if j:
                e(f"v_add_u32 {self.X(j)}, {self.X(j)}, v{g.v_t1}")
            e(f"v_lshrrev_b32 v{g.v_t1}, {LB}, {self.X(j)}")
            if j < WL - 1:
                e(f"v_and_b32 {self.X(j)}, {M}, {self.X(j)}")
            else:
                e(f"v_and_b32 {self.X(j)}, {self.X(j)}, v{g.v_t2}")
        e("s_nop 1")
        e(f"v_mov_b32_dpp v{g.v_t4}, v{g.v_t1} {self.dpp_prev} row_mask:0xf bank_mask:0xf")
        e(f"v_and_b32 v{g.v_t4}, v{g.v_t4}, v{g.v_t3}")
        e(f"v_add_u32 {self.X(0)}, {self.X(0)}, v{g.v_t4}")

    def montsq(self):
        g, e = self, self.e
        WL, H = self.WL, self.H
        e("L_montsq:")
        e("s_nop 1")
        for j in range(WL):      # helper <- a1 (its quad: lanes 4, 5 keep themselves, lanes 6, 7 take them); a1, a2 double
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xa")
        for j in range(WL):
            e(f"v_lshlrev_b32 {self.X(j)}, v{g.v_sh}, {self.X(j)}")
        self.passes("s", 0, H, True)
        self.normalize()
        # a2 += a1 a1 R^-1 (helper): lanes 2, 3 += lanes 6, 7
        self.set_exec8(0xcc)
        e("s_nop 4")
        for j in range(WL):
            e(f"v_add_u32_dpp {self.X(j)}, {self.X(j)}, {self.X(j)} row_shl:4 row_mask:0xf bank_mask:0x5")
        self.carry_X2()
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def park_addr(self):
        """v_t4 <- LDS address of this lane's parking rows: a0 in the rows of b0, a1 in those of b2, slice s at row s WL"""
        g, e = self, self.e
        row = self.NPB * 4
        e("s_nop 1")
        e(f"v_mov_b32_dpp v{g.v_t4}, v{g.v_l1mask} {self.dpp_bcast} row_mask:0xf bank_mask:0xf")    # -1 in both lanes of a1
        e(f"v_and_b32 v{g.v_t4}, {2 * self.H * row}, v{g.v_t4}")
        e(f"v_not_b32 v{g.v_t3}, v{g.v_isfirst}")
        e(f"v_and_b32 v{g.v_t3}, {self.WL * row}, v{g.v_t3}")
        e(f"v_add3_u32 v{g.v_t4}, v{g.v_t4}, v{g.v_t3}, v{g.v_aread}")

    def montmul(self):
        g, e = self, self.e
        WL, H = self.WL, self.H
        row = self.NPB * 4
        e("L_montmul:")
        e("s_nop 1")
        for j in range(WL):                                      # helper <- a0 (6 lanes down), every other lane keeps its digit
            e(f"v_mov_b32_dpp v{g.v_t5}, {self.X(j)} row_shr:6 row_mask:0xf bank_mask:0xf")
            e(f"v_bfi_b32 {self.X(j)}, v{g.v_hmask}, v{g.v_t5}, {self.X(j)}")
        # pass 1: stream b0 in the digit lanes (a0 b0 -> a1 b0 -> a2 b0 chained), stream b2 in the helper lanes (a0 b2)
        self.passes("m1", 0, 2 * H, True)
        self.normalize(dst=self.Tlo)
        # a0 and a1 still need their digits as multiplicands: park their results in the LDS rows of the finished streams
        # (a0: rows of b0, a1: rows of b2); a2 and the helper are done: pass 2 runs with them masked off
        self.park_addr()
        self.set_exec8(0x33)
        for j in range(WL):
            e(f"ds_write_b32 v{g.v_t4}, {self.Tlo(j)} offset:{j * row}")
        # pass 2: stream b1 in the lanes of a0, a1 (a0 b1 -> a1 b1 chained)
        self.passes("m2", H, H, False)
        self.normalize(dst=self.Tlo)
        # (normalize uses the scratch registers: the park address again)
        self.park_addr()
        for j in range(WL):
            e(f"ds_read_b32 {self.X(j)}, v{g.v_t4} offset:{j * row}")    # lanes of a0, a1: the parked t00 / t10
        e("s_waitcnt lgkmcnt(0)")
        # c0 = t00 | c1 = t10 + t01 | c2 = t20 + t02 + t11
        self.set_exec8(0xcc)
        e("s_nop 4")
        for j in range(WL):      # a2 <- t20 + t02 (helper, 4 lanes up)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.Tlo(j)} row_shl:4 row_mask:0xf bank_mask:0x5")
        self.set_exec8(0x33)
        e("s_nop 4")
        for j in range(WL):      # a1 += t01 (pass two of a0, 4 lanes down)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.X(j)} row_shr:4 row_mask:0xf bank_mask:0xa")
        self.set_exec8(0x3c)
        e("s_nop 4")
        for j in range(WL):      # a2 += t11 (pass two of a1, 2 lanes up)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.X(j)} row_shl:2 row_mask:0xf bank_mask:0x5")
        self.carry_X2()
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


class GenQ12(GenQ6):
    """The three-digit kernel with every digit sliced over FOUR lanes: 16 lanes per number -- one DPP row, a quad per digit:
        quad 0 = the 19-limb slices of a0 | quad 1 = those of a2 | quad 2 = those of a1 | quad 3 = the helper digit
    (digits of 76 limbs: the 74 limbs of a 2048-bit n padded, radix R_76 -- the host changes radix with one product on the way in and
    one on the way out, as for GenQ8).  GenQ6's passes, products and carries with the exchanges of a quad per digit: quotient digit and
    boundary column inside the quad, the links a0 -> a1 -> a2 by row_shr:8 and row_shl:4, the helper's copies by row_shr:4 / row_shr:12,
    the sums of the partial products by row_shl:8 / row_shr:8 / row_shl:4.  A squaring is 76 rows of 38 multiplies per lane where GenQ6
    has 74 rows of 74: for ladders modulo n^3 of at most 2 048 numbers (the verifier's W^n and x^(e0)), whose latency is the run time.
    Shared-exponent programs and limb-major per-number windows only."""

    def __init__(self, WL):
        GenQ6.__init__(self, WL)
        self.H = 4 * WL
        self.WTslot = 12 * WL
        self.WT = self.WTslot
        self.NPB = BLOCK // 16
        self.name = f"vm_asm_{WL}_160"
        self.lanes_per_number = 16
        self.lds_a = (4 * self.WLp * 4 + 15) // 16 * 16
        self.v_park = self.n_vgpr                  # LDS address of this lane's parking rows (montmul)
        self.n_vgpr += 1
        assert self.n_vgpr <= 256
        self.lds_c = self.lds_a + (self.WTslot + 1) * self.NPB * 4
        self.lds_bytes = self.lds_c + WL * 128
        assert self.lds_bytes < 65536
        self.dpp_bcast = "quad_perm:[0,0,0,0]"
        self.dpp_next = "quad_perm:[1,2,3,3]"
        self.dpp_prev = "quad_perm:[0,0,1,2]"
        self.npad = (self.H + 1) // 2 * 2

    def set_exec16(self, mask16):
        w = mask16 | (mask16 << 16)
        self.e(f"s_mov_b32 exec_lo, {hex(w)}")
        self.e(f"s_mov_b32 exec_hi, {hex(w)}")

    def mask_digit_lanes(self, on):
        if on:
            self.set_exec16(0x0fff)
        else:
            self.e("s_mov_b64 exec, -1")

    def prologue(self):
        g, e = self, self.e
        WL, NPB, H = self.WL, self.NPB, self.H
        row = NPB * 4
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 15, v0")                 # k16
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the constants table
        e(f"v_lshrrev_b32 v{g.v_t2}, 4, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        # slot position of the lane: digit dd = (0, 2, 1, 2)[quad] (the helper aliases digit two; its stores are masked), slice s
        e(f"v_lshrrev_b32 v{g.v_t4}, 2, v{g.v_t1}")       # q = k16 >> 2
        e(f"v_and_b32 v{g.v_t5}, 1, v{g.v_t4}")           # t = q & 1
        e(f"v_xor_b32 v{g.v_p0}, 1, v{g.v_t5}")           # t ^ 1
        e(f"v_lshrrev_b32 v{g.v_p0 + 1}, 1, v{g.v_t4}")   # q >> 1
        e(f"v_and_b32 v{g.v_p0}, v{g.v_p0}, v{g.v_p0 + 1}")
        e(f"v_lshl_or_b32 v{g.v_t4}, v{g.v_t5}, 1, v{g.v_p0}")   # dd
        e(f"v_and_b32 v{g.v_t5}, 3, v{g.v_t1}")           # s
        e(f"v_lshl_or_b32 v{g.v_t4}, v{g.v_t4}, 2, v{g.v_t5}")   # kk = 4 dd + s
        e(f"s_mul_i32 s{g.s_t1}, s15, {WL}")
        e(f"v_mul_lo_u32 v{g.v_p0}, v{g.v_t4}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_p0}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_add_u32 v{g.v_aread}, {self.lds_a}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_p0}, {WL * NPB * 4}, v{g.v_t4}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_p0}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {WL * 4}, v{g.v_t4}")
        e(f"v_mul_u32_u24 v{g.v_nbase}, {self.WLp * 4}, v{g.v_t5}")
        # parking rows of a product's first pass: a0 in the rows of b0, a1 in those of b2, slice s at row s WL
        e(f"v_mul_u32_u24 v{g.v_park}, {WL * row}, v{g.v_t5}")
        e(f"v_add_u32 v{g.v_park}, v{g.v_park}, v{g.v_aread}")
        e(f"v_lshrrev_b32 v{g.v_p0}, 3, v{g.v_t1}")        # 1 in quads 2, 3
        e(f"v_mul_u32_u24 v{g.v_p0}, {2 * H * row}, v{g.v_p0}")
        e(f"v_add_u32 v{g.v_park}, v{g.v_park}, v{g.v_p0}")
        e(f"v_cmp_eq_u32 vcc, 0, v{g.v_t5}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_isfirst}, 0, -1, vcc")
        e(f"v_cmp_ne_u32 vcc, 3, v{g.v_t5}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_notlast}, 0, -1, vcc")
        for val, reg in ((8, g.v_l1mask), (4, g.v_l2mask)):
            e(f"v_cmp_eq_u32 vcc, {val}, v{g.v_t1}")
            e("s_nop 1")
            e(f"v_cndmask_b32 v{reg}, 0, -1, vcc")
        e(f"v_cmp_le_u32 vcc, 12, v{g.v_t1}")
        e("s_nop 1")
        e(f"v_cndmask_b32 v{g.v_hmask}, 0, -1, vcc")
        e(f"v_add_u32 v{g.v_sh}, 4, v{g.v_t1}")
        e(f"v_bfe_u32 v{g.v_sh}, v{g.v_sh}, 3, 1")         # 1 in the lanes of a2 and a1 (k16 = 4 .. 11): they double their multiplicand
        # modulus n (4 slices) -> LDS -> this lane's slice in VGPRs
        e(f"v_lshlrev_b32 v{g.v_t3}, 2, v0")
        e(f"v_cmp_gt_u32 vcc, {WL}, v0")
        e("s_nop 1")
        e("s_and_saveexec_b64 s[96:97], vcc")
        for sgi in range(4):
            e(f"global_load_dword v{g.v_p1}, v{g.v_t3}, s[6:7] offset:{sgi * WL * 4}")
            e("s_waitcnt vmcnt(0)")
            e(f"ds_write_b32 v{g.v_t3}, v{g.v_p1} offset:{sgi * self.WLp * 4}")
        # constants table: thread t < WL writes row t = (0 x 4 | C2 slices | C1 slices | 0 x 4), zero-extended to 64 bits
        e(f"v_lshlrev_b32 v{g.v_t4}, 5, v{g.v_t3}")        # t * 128
        e(f"v_lshlrev_b32 v{g.v_t5}, 1, v{g.v_t3}")        # t * 8: the pair (C1_t, C2_t)
        e(f"v_mov_b32 v{g.v_y0}, 0")
        e(f"v_mov_b32 v{g.v_y0 + 1}, 0")
        for k in (0, 1, 2, 3, 12, 13, 14, 15):
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * k}")
        for sgi in range(4):
            e(f"global_load_dwordx2 {self.P(g.v_p0)}, v{g.v_t5}, s[6:7] offset:{4 * self.npad + 8 * WL * sgi}")
            e("s_waitcnt vmcnt(0)")
            e(f"v_mov_b32 v{g.v_y0}, v{g.v_p0 + 1}")       # (C2, 0)
            e(f"v_mov_b32 v{g.v_p0 + 1}, 0")               # (C1, 0)
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_p0)} offset:{self.lds_c + 8 * (8 + sgi)}")
            e(f"ds_write_b64 v{g.v_t4}, {self.P(g.v_y0)} offset:{self.lds_c + 8 * (4 + sgi)}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_mov_b64 exec, s[96:97]")
        e("s_barrier")
        for j in range(WL):
            e(f"ds_read_b32 v{g.v_N + j}, v{g.v_nbase} offset:{4 * j}")
        e("s_waitcnt lgkmcnt(0)")
        for j in range(WL):
            e(f"v_mov_b32 {self.X(j)}, 0")
        self.init_row_regs()

    def chain_steps(self, link2):
        g = self
        m = f"v{g.v_m}"
        chain = [f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        for ctrl, mask in [("row_shr:8", g.v_l1mask)] + ([("row_shl:4", g.v_l2mask)] if link2 else []):
            chain += [f"v_and_b32_dpp v{g.v_d}, {m}, v{g.v_mask28} {ctrl} row_mask:0xf bank_mask:0xf",
                      f"v_mad_i64_i32 {self.T(0)}, vcc, v{g.v_d}, v{mask}, {self.T(0)}",
                      f"v_mul_lo_u32 {m}, {self.Tlo(0)}, s14"]
        chain.append(f"v_and_b32_dpp {m}, {m}, v{g.v_mask28} {self.dpp_bcast} row_mask:0xf bank_mask:0xf")
        return chain

    def passes(self, tag, off_d, off_h, link2):
        g, e = self, self.e
        WL, H = self.WL, self.H
        row = self.NPB * 4
        for j in range(WL):                                      # accumulators <- (0 | C2 | C1 | 0) by quad
            e(f"ds_read_b64 {self.T(j)}, v{g.v_caddr} offset:{self.lds_c + 128 * j}")
        e(f"v_and_b32 v{g.v_t1}, {(off_h - off_d) * row}, v{g.v_hmask}")
        e(f"v_add_u32 v{g.v_arow}, v{g.v_aread}, v{g.v_t1}")
        if off_d:
            e(f"v_add_u32 v{g.v_arow}, {off_d * row}, v{g.v_arow}")
        self.read_pair(g.v_pa, g.v_arow, 0)
        self.row_loop(tag, link2, False)

    def montsq(self):
        g, e = self, self.e
        WL, H = self.WL, self.H
        e("L_montsq:")
        e("s_nop 1")
        for j in range(WL):      # helper <- a1 (quad 3 takes quad 2's slices); a1, a2 double
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} row_shr:4 row_mask:0xf bank_mask:0x8")
        for j in range(WL):
            e(f"v_lshlrev_b32 {self.X(j)}, v{g.v_sh}, {self.X(j)}")
        self.passes("s", 0, H, True)
        self.normalize()
        # a2 += a1 a1 R^-1 (helper): quad 1 += quad 3
        self.set_exec16(0xf0f0)
        e("s_nop 4")
        for j in range(WL):
            e(f"v_add_u32_dpp {self.X(j)}, {self.X(j)}, {self.X(j)} row_shl:8 row_mask:0xf bank_mask:0x2")
        self.carry_X2()
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def montmul(self):
        g, e = self, self.e
        WL, H = self.WL, self.H
        row = self.NPB * 4
        e("L_montmul:")
        e("s_nop 1")
        for j in range(WL):                                      # helper <- a0 (12 lanes down), every other lane keeps its digit
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} row_shr:12 row_mask:0xf bank_mask:0x8")
        # pass 1: stream b0 in the digit lanes (a0 b0 -> a1 b0 -> a2 b0 chained), stream b2 in the helper lanes (a0 b2)
        self.passes("m1", 0, 2 * H, True)
        self.normalize(dst=self.Tlo)
        # a0 and a1 still need their digits as multiplicands: park their results in the LDS rows of the finished streams
        # (a0: rows of b0, a1: rows of b2); a2 and the helper are done: pass 2 runs with them masked off
        self.set_exec16(0x0f0f)
        for j in range(WL):
            e(f"ds_write_b32 v{g.v_park}, {self.Tlo(j)} offset:{j * row}")
        # pass 2: stream b1 in the lanes of a0, a1 (a0 b1 -> a1 b1 chained)
        self.passes("m2", H, H, False)
        self.normalize(dst=self.Tlo)
        for j in range(WL):
            e(f"ds_read_b32 {self.X(j)}, v{g.v_park} offset:{j * row}")    # lanes of a0, a1: the parked t00 / t10
        e("s_waitcnt lgkmcnt(0)")
        # c0 = t00 | c1 = t10 + t01 | c2 = t20 + t02 + t11
        self.set_exec16(0xf0f0)
        e("s_nop 4")
        for j in range(WL):      # a2 <- t20 + t02 (helper, 8 lanes up)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.Tlo(j)} row_shl:8 row_mask:0xf bank_mask:0x2")
        self.set_exec16(0x0f0f)
        e("s_nop 4")
        for j in range(WL):      # a1 += t01 (pass two of a0, 8 lanes down)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.X(j)} row_shr:8 row_mask:0xf bank_mask:0x4")
        self.set_exec16(0x0ff0)
        e("s_nop 4")
        for j in range(WL):      # a2 += t11 (pass two of a1, 4 lanes up)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.X(j)} row_shl:4 row_mask:0xf bank_mask:0x2")
        self.carry_X2()
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")


class GenQ3(LaneRows, Gen):
    """Three-digit kernel for moduli N = n^3 with n PUBLIC (level-two Encrypt / ConstMult / NestedRandomize, the DDLEQ
    equations): the residue y R mod n^3 = a0 + a1 n + a2 n^2 (R = 2^(28 H), H = limbs of n, digits lazily reduced mod n)
    lives in the lanes of a quad -- lane d holds digit a_d, lane 3 is a helper -- and all four lanes run the single-lane
    Montgomery row modulo n (n in SGPRs) on their own block a_i * b_j of the product
        y z = B00 + (B10 + B01) n + (B20 + B11 + B02) n^2   (mod n^3),    B_ij = a_i b_j.
    A block is reduced on its own; what makes the digits exact modulo n^3 (not just modulo n) is that the QUOTIENT DIGITS
    of a block's reduction are subtracted from a block of the next digit in the same row -- one DPP move per hop, as in
    GenQ:   B00 -> m00 -> (B10 - m00 + C1) -> m10 -> (B20 - m10 + C2),      (B01) -> m01 -> (B11 - m01 + C1),
    with C1 = k1 n a multiple of n whose limbs all exceed 2^28 (so the seeds stay limb-wise non-negative) and C2 = -k1 mod n
    of the same shape (the -C1 n of digit one reappears as -k1 n^2 in digit two).
      squaring  ONE pass of H rows: lane 0 a0 a0 | lane 1 2 a0 a1 | lane 2 2 a0 a2 | lane 3 a1 a1 (-> added to digit 2):
                8 H^2 multiplies per number where the 4-lane 3H-limb kernel needs 18 H^2;
      product   two passes: (a0 b0 | a1 b0 | a2 b0 | a0 b2), then (a0 b1 | a1 b1) with lanes 2, 3 masked off: 12 H^2
                multiplies in 16 H^2 issue slots, against 18 H^2.
    Multiplier streams come from the LDS column of the staged operand (each lane its own row offset), multiplicand
    vectors are the lanes' own digits (lane 3 copies the one it needs by DPP).  No inter-wave traffic; one barrier, after
    the block has copied the constants to LDS.
    The constants enter once per pass, not once per row: the accumulators of a pass START at (0 | C1 | C2 | 0) by lane
    (limb j of the constant is the initial value of column j, which is all the per-row addition ever did), read from a
    small LDS table [H][4 lanes] of zero-extended limbs.  A hop is then DPP move, T0 -= m (a signed multiply-add with the
    lane's -1 / 0 mask), quotient digit.
    Slot layout: 3H limbs -- a0 | a1 | a2.  `nmod` points at n (H limbs, padded to an even count) followed by the
    interleaved pairs (C1_i, C2_i)."""

    def __init__(self, H=74):
        Gen.__init__(self, H, 4)
        assert 3 * H + 3 <= 255
        self.H = H
        self.name = f"vm_asm_{H}_48"
        self.WT = 3 * H                 # slot / constant width used by the dispatcher's addressing
        self.NPB = 64
        self.n_sgpr = True
        self.n_vreg = False
        self.flush = False
        self.sq_rows = True
        self.sq_rows_k = False
        self.lds_a = 0
        self.lds_c = (3 * H + 1) * self.NPB * 4          # constants table: [H][4] zero-extended limbs (0 | C1_j | C2_j | 0)
        self.lds_bytes = self.lds_c + H * 32
        assert self.lds_c + H * 32 < 65536                # LDS instruction offsets are 16 bits
        self.vX = 2 * H
        e = 3 * H
        for nm in ["m", "t1", "sh", "l1mask", "l2mask", "l3mask", "mask28"]:
            setattr(self, "v_" + nm, e)
            e += 1
        e = (e + 1) // 2 * 2
        self.v_pa, self.v_pb = e, e + 2      # multiplier pairs: two rows are read at a time
        e += 4
        self.v_y0 = e
        e += 2
        self.v_d = e          # pair (adjustment, 0)
        e += 2
        for nm in ["goff", "aread", "awrite", "arow", "t2", "t3", "t4", "koff", "caddr", "tgoff"]:
            setattr(self, "v_" + nm, e)
            e += 1
        self.v_addr = self.v_arow
        e = (e + 1) // 2 * 2
        self.v_p0 = e
        e += 2
        self.v_p1 = e
        self.v_c = e
        e += 2
        # ONE-pass product (H = 37, 55): lanes 1, 2 keep a copy of the digit below their own (H more registers, taken by DPP) and
        # run two multiplier streams per row -- b0 against their own digit, b1 against the copy -- into the same accumulators:
        #   lane 0: a0 b0 | lane 1: a1 b0 + a0 b1 - m0 + C1 | lane 2: a2 b0 + a1 b1 - m1 + C2 | helper: a0 b2 (second stream: zeros)
        # One Montgomery reduction per digit, whose quotient digits go to the next digit as before: the same value modulo n^3
        # in H rows of 3H multiplies instead of two passes of H rows of 2H.  H = 74 has no registers for it.
        self.merged = (e + H + 6 <= 256)
        if self.merged:
            self.v_pa2, self.v_pb2 = e, e + 2            # (e is even here)
            e += 4
            self.vY = e
            e += H
            self.v_arow2, self.v_bump2 = e, e + 1
            e += 2
            self.lds_z = self.lds_bytes                  # 2 KB of zeros: the second stream of lanes 0 and 3 (a paired read
            self.lds_bytes += 2048                       # reaches up to six rows past the pointer)
            assert self.lds_bytes < 65536
        self.n_vgpr = e
        assert e <= 256, e
        self.s_coff = 99
        self.number_major_tables = True
        self.npad = (H + 1) // 2 * 2    # n occupies an even number of words so that the pairs are 8-byte aligned

    def hops(self, link2):
        return [("quad_perm:[0,0,1,2]", self.v_l1mask)] + ([("quad_perm:[0,0,1,2]", self.v_l2mask)] if link2 else [])

    def wide_chunks(self):
        """(first limb, dwords) pieces covering the H limbs of a digit with the widest loads / stores"""
        out, j = [], 0
        while j < self.H:
            n = 4 if self.H - j >= 4 else 2 if self.H - j >= 2 else 1
            out.append((j, n))
            j += n
        return out

    def number_major_ops(self, St):
        """Tables that are gathered per number (the windows of per-number exponents) are stored NUMBER-major inside their
        slots -- [number][3H limbs] instead of [3H limbs][number] -- by STORET, and read back by MULV7: a lane's digit is 4 H
        contiguous bytes, fetched with 16-byte loads that use every byte of the lines they touch.  (Limb-major, a gather
        reads one dword per line: neighbouring numbers want different table entries.)"""
        g, e = self, self.e
        H = self.H
        sfx = {4: "x4", 2: "x2", 1: ""}
        e("L_storet:")
        self.slot_base()
        self.mask_digit_lanes(True)
        for j, n in self.wide_chunks():
            src = self.X(j) if n == 1 else f"v[{g.vX + j}:{g.vX + j + n - 1}]"
            e(f"global_store_dword{sfx[n]} v{g.v_tgoff}, {src}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
        e("s_waitcnt vmcnt(0)")
        self.mask_digit_lanes(False)
        e("s_branch L_next")

        for lbl, per_word, wbits in (("L_mulv7", 4, 7), ("L_mulvt5", 5, 5)):
            # table entry = aux + the window `arg` of this number's own exponent: 7 bits, 4 per 28-bit limb (MULV7), or -- batches
            # whose 128-entry tables would not fit the 32-bit gather offsets -- 5 bits, 5 per 25-bit word of the repacked exponent (MULVT5)
            e(f"{lbl}:")
            e(f"s_mul_hi_u32 s{g.s_t1}, s17, {((1 << 32) + per_word - 1) // per_word}")   # q = arg / per_word
            e(f"s_mul_i32 s98, s{g.s_t1}, {per_word}")
            e("s_sub_u32 s98, s17, s98")
            e(f"s_mul_i32 s98, s98, {wbits}")                                  # shift = wbits (arg % per_word)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s{g.s_t1}, s3")                 # digits + q * nb*4
            e(f"s_mul_i32 s{g.s_sbase}, s{g.s_t1}, s3")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s12")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s13")
            e(f"v_subrev_u32 v{g.v_t2}, {self.lds_a}, v{g.v_aread}")           # gl*4
            e(f"s_mul_i32 s{g.s_t0}, s2, {self.NPB * 4}")
            e(f"v_add_u32 v{g.v_t2}, s{g.s_t0}, v{g.v_t2}")                    # g*4
            e(f"global_load_dword v{g.v_t3}, v{g.v_t2}, s[{g.s_sbase}:{g.s_sbase + 1}]")
            e("s_waitcnt vmcnt(0)")
            e(f"v_lshrrev_b32 v{g.v_t3}, s98, v{g.v_t3}")
            e(f"v_and_b32 v{g.v_t3}, {(1 << wbits) - 1}, v{g.v_t3}")           # digit
            e(f"s_mul_i32 s{g.s_t0}, s3, {self.WT}")                           # slot stride in bytes
            e(f"v_mul_lo_u32 v{g.v_t3}, v{g.v_t3}, s{g.s_t0}")                  # digit * stride (host guarantees < 2^32)
            e("s_bfe_u32 s17, s16, 0x160008")                                  # aux = first table slot (bits 8..29)
            e(f"s_mul_hi_u32 s{g.s_sbase + 1}, s17, s{g.s_t0}")
            e(f"s_mul_i32 s{g.s_sbase}, s17, s{g.s_t0}")
            e(f"s_add_u32 s{g.s_sbase}, s{g.s_sbase}, s10")
            e(f"s_addc_u32 s{g.s_sbase + 1}, s{g.s_sbase + 1}, s11")
            e(f"v_add_u32 v{g.v_addr}, v{g.v_t3}, v{g.v_tgoff}")
            for j, n in self.wide_chunks():
                dst = St[j] if n == 1 else f"v[{j}:{j + n - 1}]"
                e(f"global_load_dword{sfx[n]} {dst}, v{g.v_addr}, s[{g.s_sbase}:{g.s_sbase + 1}] offset:{4 * j}")
            e("s_waitcnt vmcnt(0)")
            self.stage_to_lds(St)
            e("s_branch L_montmul")

    def set_exec(self, mask4):
        """exec <- the lanes of every quad selected by the 4-bit mask"""
        w = sum(mask4 << (4 * i) for i in range(8))
        self.e(f"s_mov_b32 exec_lo, {hex(w)}")
        self.e(f"s_mov_b32 exec_hi, {hex(w)}")

    def mask_digit_lanes(self, on):
        if on:
            self.set_exec(0x7)
        else:
            self.e("s_mov_b64 exec, -1")

    def prologue(self):
        g, e = self, self.e
        H, NPB = self.H, self.NPB
        e(f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
        e(".text")
        e(f".globl {self.name}")
        e(".p2align 8")
        e(f".type {self.name},@function")
        e(f"{self.name}:")
        self.select_segment()
        e("s_load_dwordx8 s[4:11], s[0:1], 0x0")
        e("s_load_dwordx4 s[12:15], s[0:1], 0x20")
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshl_b32 s3, s15, 2")
        e(f"v_and_b32 v{g.v_t1}, 3, v0")                  # k: 0..2 digit lanes, 3 helper
        e(f"v_lshrrev_b32 v{g.v_t2}, 2, v0")              # gl
        e(f"s_mul_i32 s{g.s_t0}, s2, {NPB}")
        e(f"v_add_u32 v{g.v_t3}, s{g.s_t0}, v{g.v_t2}")   # g
        e(f"v_min_u32 v{g.v_t4}, 2, v{g.v_t1}")           # the helper lane aliases digit 2 for addressing (its stores are masked)
        e(f"s_mul_i32 s{g.s_t1}, s15, {H}")
        e(f"v_mul_lo_u32 v{g.v_koff}, v{g.v_t4}, s{g.s_t1}")
        e(f"v_add_lshl_u32 v{g.v_goff}, v{g.v_koff}, v{g.v_t3}, 2")
        e(f"v_lshlrev_b32 v{g.v_aread}, 2, v{g.v_t2}")
        e(f"v_mul_u32_u24 v{g.v_awrite}, {H * NPB * 4}, v{g.v_t4}")
        e(f"v_add_u32 v{g.v_awrite}, v{g.v_awrite}, v{g.v_aread}")
        e(f"v_mul_u32_u24 v{g.v_koff}, {H * 4}, v{g.v_t4}")
        e(f"v_mul_u32_u24 v{g.v_tgoff}, {3 * H * 4}, v{g.v_t3}")      # number-major table slots: this lane's digit of number g
        e(f"v_add_u32 v{g.v_tgoff}, v{g.v_tgoff}, v{g.v_koff}")
        for lane, reg in ((1, g.v_l1mask), (2, g.v_l2mask), (3, g.v_l3mask)):
            e(f"v_cmp_eq_u32 vcc, {lane}, v{g.v_t1}")
            e("s_nop 1")
            e(f"v_cndmask_b32 v{reg}, 0, -1, vcc")
        e(f"v_mov_b32 v{g.v_mask28}, {hex(MASK)}")
        off, s, rem = 0, self.s_N, H
        while rem > 0:
            for cnt in (16, 8, 4, 2, 1):
                align = 4 if cnt >= 4 else cnt
                if cnt <= rem and s % align == 0:
                    if cnt == 1:
                        e(f"s_load_dword s{s}, s[6:7], {hex(off)}")
                    else:
                        e(f"s_load_dwordx{cnt} s[{s}:{s + cnt - 1}], s[6:7], {hex(off)}")
                    off += 4 * cnt
                    s += cnt
                    rem -= cnt
                    break
            else:
                raise RuntimeError("cannot tile the modulus into SGPR loads")
        e("s_waitcnt lgkmcnt(0)")
        # constants table in LDS: thread t < H copies the pair (C1_t, C2_t) as the four zero-extended limbs of row t
        e(f"v_lshlrev_b32 v{g.v_caddr}, 3, v{g.v_t1}")    # this lane's column of the table: (lane & 3) * 8
        e(f"v_cmp_gt_u32 vcc, {H}, v0")
        e("s_and_saveexec_b64 s[0:1], vcc")
        e("v_lshlrev_b32 v0, 3, v0")                      # t * 8
        e(f"global_load_dwordx2 v[2:3], v0, s[6:7] offset:{4 * self.npad}")
        e("v_lshlrev_b32 v0, 2, v0")                      # t * 32
        e("v_mov_b32 v4, 0")
        e("v_mov_b32 v5, 0")
        e("s_waitcnt vmcnt(0)")
        e("v_mov_b32 v6, v3")                             # (C2_t, 0)
        e("v_mov_b32 v7, 0")
        e("v_mov_b32 v3, 0")                              # (C1_t, 0)
        e(f"ds_write_b64 v0, v[4:5] offset:{self.lds_c}")
        e(f"ds_write_b64 v0, v[2:3] offset:{self.lds_c + 8}")
        e(f"ds_write_b64 v0, v[6:7] offset:{self.lds_c + 16}")
        e(f"ds_write_b64 v0, v[4:5] offset:{self.lds_c + 24}")
        e("s_mov_b64 exec, s[0:1]")
        if self.merged:
            e(f"v_lshlrev_b32 v{g.v_t3}, 2, v{g.v_t2}")        # gl * 4 ... every thread zeroes one word of the zero rows
            e(f"v_lshl_add_u32 v{g.v_t3}, v{g.v_t1}, 8, v{g.v_t3}")   # + (lane & 3) * 256
            e("v_mov_b32 v2, 0")
            e(f"ds_write_b32 v{g.v_t3}, v2 offset:{self.lds_z}")
            e(f"ds_write_b32 v{g.v_t3}, v2 offset:{self.lds_z + 1024}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        for j in range(H):
            e(f"v_mov_b32 {self.X(j)}, 0")

    def passes(self, tag, off012, off3, link2, use_sh, two_streams=False):
        """H rows: T <- (multiplier stream) * X * R^-1 with the quotient links.  off012 / off3: first row of the stream
        read by the digit lanes / by the helper lane."""
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        for j in range(H):                                   # accumulators <- (0 | C1_j | C2_j | 0) by lane
            e(f"ds_read_b64 {self.T(j)}, v{g.v_caddr} offset:{self.lds_c + 32 * j}")
        e(f"v_and_b32 v{g.v_t1}, {(off3 - off012) * row}, v{g.v_l3mask}")
        e(f"v_add_u32 v{g.v_arow}, v{g.v_aread}, v{g.v_t1}")
        if off012:
            e(f"v_add_u32 v{g.v_arow}, {off012 * row}, v{g.v_arow}")
        if two_streams:
            # stream two: rows H.. of the a column (b1) in lanes 1, 2; the zero rows in lanes 0 and 3 (pointer not advanced)
            e(f"v_or_b32 v{g.v_t2}, v{g.v_l1mask}, v{g.v_l2mask}")
            e(f"v_add_u32 v{g.v_arow2}, {H * row}, v{g.v_aread}")
            e(f"v_mov_b32 v{g.v_t1}, {self.lds_z}")
            e(f"v_xor_b32 v{g.v_t1}, v{g.v_t1}, v{g.v_arow2}")
            e(f"v_and_b32 v{g.v_t1}, v{g.v_t1}, v{g.v_t2}")
            e(f"v_xor_b32 v{g.v_arow2}, {self.lds_z}, v{g.v_t1}")       # lanes 1, 2: aread + H rows; lanes 0, 3: lds_z
            e(f"v_and_b32 v{g.v_bump2}, {row}, v{g.v_t2}")
        self.rows(tag, link2, use_sh, two_streams)

    def carry_T(self, to_x=False):
        """sequential carry through the accumulators: canonical limb j ends up in Tlo(j) (to_x: in X(j), the multiplicand
        being dead by then); the top limb keeps its excess"""
        g, e = self, self.e
        H = self.H
        c = self.P(g.v_c)
        for j in range(H):
            if j:
                e(f"v_lshl_add_u64 {self.T(j)}, {self.T(j)}, 0, {c}")
            dst = self.X(j) if to_x else self.Tlo(j)
            if j < H - 1:
                e(f"v_lshrrev_b64 {c}, {LB}, {self.T(j)}")
                e(f"v_and_b32 {dst}, {hex(MASK)}, {self.Tlo(j)}")
            elif to_x:
                e(f"v_mov_b32 {dst}, {self.Tlo(j)}")

    def carry_X(self, extra=None):
        """lazy limbs (sums of up to three canonical limbs) -> canonical again; the top limb keeps the excess.
        extra(j): a register added to limb j on the way (saves the separate addition)"""
        g, e = self, self.e
        H = self.H
        for j in range(H):
            if j and extra:
                e(f"v_add3_u32 {self.X(j)}, {self.X(j)}, {extra(j)}, v{g.v_t1}")
            elif j:
                e(f"v_add_u32 {self.X(j)}, {self.X(j)}, v{g.v_t1}")
            elif extra:
                e(f"v_add_u32 {self.X(j)}, {self.X(j)}, {extra(j)}")
            if j < H - 1:
                e(f"v_lshrrev_b32 v{g.v_t1}, {LB}, {self.X(j)}")
                e(f"v_and_b32 {self.X(j)}, {hex(MASK)}, {self.X(j)}")

    def montsq(self):
        g, e = self, self.e
        H = self.H
        e("L_montsq:")
        e("s_nop 1")
        for j in range(H):                                   # helper lane <- a1 (lanes 0..2 keep their own digit)
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf")
        e(f"v_or_b32 v{g.v_sh}, v{g.v_l1mask}, v{g.v_l2mask}")
        e(f"v_and_b32 v{g.v_sh}, 1, v{g.v_sh}")              # lanes 1, 2 double the multiplier: 2 a0 a1, 2 a0 a2
        self.passes("s", 0, H, True, True)
        self.carry_T(to_x=True)
        # digit 2 += a1 a1 R^-1 (helper lane): one DPP addition per limb with only lanes 2, 3 enabled (the source lane of a
        # DPP operand has to be active; what the helper lane computes for itself is never read)
        self.set_exec(0xc)
        e("s_nop 4")
        for j in range(H):
            e(f"v_add_u32_dpp {self.X(j)}, {self.X(j)}, {self.X(j)} quad_perm:[0,1,3,3] row_mask:0xf bank_mask:0xf")
        self.carry_X()                                       # lanes 0, 1 are canonical already and stay as they are
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def montmul_merged(self):
        g, e = self, self.e
        H = self.H
        e("L_montmul:")
        e("s_nop 1")
        for j in range(H):                                   # lanes 1, 2 <- the digit below their own (a0, a1)
            e(f"v_mov_b32_dpp v{g.vY + j}, {self.X(j)} quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf")
        for j in range(H):                                   # helper lane <- a0
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} quad_perm:[0,1,2,0] row_mask:0xf bank_mask:0xf")
        # one pass: stream b0 in the digit lanes (a0 b0 -> a1 b0 -> a2 b0 chained) and b2 in the helper lane (a0 b2); stream b1
        # against the copies in lanes 1, 2 (a0 b1, a1 b1), zeros elsewhere
        self.passes("m", 0, 2 * H, True, False, two_streams=True)
        self.carry_T(to_x=True)
        # digit 2 += a0 b2 R^-1 (helper lane), as in the squaring
        self.set_exec(0xc)
        e("s_nop 4")
        for j in range(H):
            e(f"v_add_u32_dpp {self.X(j)}, {self.X(j)}, {self.X(j)} quad_perm:[0,1,3,3] row_mask:0xf bank_mask:0xf")
        self.carry_X()
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def montmul(self):
        if self.merged:
            return self.montmul_merged()
        g, e = self, self.e
        H = self.H
        row = self.NPB * 4
        e("L_montmul:")
        e("s_nop 1")
        for j in range(H):                                   # helper lane <- a0
            e(f"v_mov_b32_dpp {self.X(j)}, {self.X(j)} quad_perm:[0,1,2,0] row_mask:0xf bank_mask:0xf")
        # pass 1: stream b0 in the digit lanes (a0 b0 -> a1 b0 -> a2 b0 chained), stream b2 in the helper lane (a0 b2)
        self.passes("m1", 0, 2 * H, True, False)
        self.carry_T()
        # lanes 0, 1 still need their digits as multiplicands: park their results in the LDS rows of the streams that are
        # finished (lane 0: rows of b0, lane 1: rows of b2); lanes 2, 3 are done: pass 2 runs with them masked off, so
        # their results simply stay in Tlo
        e(f"v_and_b32 v{g.v_t4}, {2 * H * row}, v{g.v_l1mask}")
        e(f"v_add_u32 v{g.v_t4}, v{g.v_t4}, v{g.v_aread}")
        self.set_exec(0x3)
        for j in range(H):
            e(f"ds_write_b32 v{g.v_t4}, {self.Tlo(j)} offset:{j * row}")
        # pass 2: stream b1 in lanes 0, 1 (a0 b1 -> a1 b1 chained)
        self.passes("m2", H, H, False, False)
        self.carry_T()
        # c0 = t00 | c1 = t10 + t01 | c2 = t20 + t02 + t11
        for j in range(H):
            e(f"ds_read_b32 {self.X(j)}, v{g.v_t4} offset:{j * row}")    # lanes 0, 1: the parked t00 / t10
        e("s_mov_b64 exec, -1")
        e("s_nop 4")
        for j in range(H):    # lane 1 <- t01, lane 2 <- t11, kept in the dead half of T(j)
            e(f"v_mov_b32_dpp {self.Thi(j)}, {self.Tlo(j)} quad_perm:[0,0,1,3] row_mask:0xf bank_mask:0xf")
        e("s_waitcnt lgkmcnt(0)")
        self.set_exec(0xc)
        e("s_nop 4")
        for j in range(H):    # lane 2 <- t20 + t02 (the helper lane's own sum is never read)
            e(f"v_add_u32_dpp {self.X(j)}, {self.Tlo(j)}, {self.Tlo(j)} quad_perm:[0,1,3,3] row_mask:0xf bank_mask:0xf")
        self.set_exec(0x6)    # lane 0 (t00) is canonical and final
        self.carry_X(extra=lambda j: self.Thi(j))
        e("s_mov_b64 exec, -1")
        e("s_branch L_next")

    def generate(self):
        self.prologue()
        self.dispatcher()
        self.montmul()
        self.montsq()
        self.epilogue()
        return "\n".join(self.lines) + "\n"


SHAPES = [(74, 1), (37, 1), (55, 1), (55, 2), (74, 2), (55, 4), (74, 4), (42, 8), (37, 2), (37, 4), (10, 4), (37, 16), (55, 16), (74, 32), (55, 32), (37, 32), (37, 64), (19, 96), (10, 96), (10, 128), (74, 48), (37, 48), (55, 48), (55, 112), (37, 112), (19, 112), (19, 160)]
PAIR = {(37, 16), (55, 16)}  # (H, 16): the pair kernel for N = p^2 with H-limb p (GenP); 16 is a tag, not a lane count
PAIR4 = {(37, 64)}          # (WL, 64): GenQ4, the two digits of GenQ(2 WL) sliced over two lanes each
PAIR8 = {(19, 96), (10, 96)}
PAIR16 = {(10, 128)}        # (WL, 128): GenQ16, the two digits sliced over eight lanes each (80-limb digits)          # (WL, 96): GenQ8, the two digits sliced over four lanes each (76-limb digits)
TRIPLE2 = {(55, 112), (37, 112), (19, 112)}
TRIPLE4 = {(19, 160)}      # (WL, 160): GenQ12, three digits of 4 WL limbs, four lanes each (+ four helper lanes): a DPP row per number   # (WL, 112): GenQ6, three digits of 2 WL limbs, two lanes each (+ two helper lanes)
PAIR2 = {(74, 32), (55, 32), (37, 32)}          # (H, 32): the two-lane pair kernel for N = n^2 with H-limb n (GenQ)
WAVE_SLICED = {(74, 2), (55, 2)}     # shapes whose slices live in different waves (GenW) instead of neighbouring lanes (Gen)
TRIPLE = {(74, 48), (37, 48), (55, 48)}        # (H, 48): GenQ3, residues modulo n^3 as three base-n digits in the lanes of a quad


def make_gen(wl, k):
    if (wl, k) in PAIR:
        return GenP2(wl) if wl > 37 else GenP(wl)
    if (wl, k) in PAIR2:
        return GenQ(wl)
    if (wl, k) in PAIR4:
        return GenQ4(wl)
    if (wl, k) in PAIR8:
        return GenQ8(wl)
    if (wl, k) in PAIR16:
        return GenQ16(wl)
    if (wl, k) in TRIPLE4:
        return GenQ12(wl)
    if (wl, k) in TRIPLE2:
        return GenQ6(wl)
    if (wl, k) in TRIPLE:
        return GenQ3(wl)
    if (wl, k) in WAVE_SLICED:
        return GenW(wl, k)
    if (wl, k) == (37, 1) and os.environ.get("PGPU_GEN_LOOPED37", "0") == "1":      # A/B builds: the looped rows of Gen with the same tables
        g = Gen(wl, k)
        g.nm4_tables = True
        return g
    if (wl, k) == (37, 1):
        # the one-lane 37-limb kernel runs the key holder's ladders modulo the primes of a 2048-bit key with per-number exponents
        # (struct_pow_n3): unrolled products, number-major window tables (a limb-major gather reads one dword per 32-byte sector)
        return GenM(wl)
    if (wl, k) == (10, 4):
        # the same 37-limb primes as 40-limb moduli in four lanes per number: ladders modulo the primes of batches too small to fill
        # the chip with one lane per number (plan::prime_lanes) -- a product is 40 rows of 20 multiplies where the one-lane kernel
        # has 2 053 - 2 738 in a row, and the ladder's latency is the run time there
        if os.environ.get("PGPU_GEN_GENERIC10", "0") == "1":       # A/B builds: the generic four-lane rows of Gen
            g = Gen(wl, k)
            g.nm4_tables = True
            return g
        return GenS4(wl)
    return Gen(wl, k)


if __name__ == "__main__":
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
    for wl, k in SHAPES:
        g = make_gen(wl, k)
        with open(f"{out_dir}/vm_asm_{wl}_{k}.s", "w") as f:
            f.write(g.generate())
        print(f"vm_asm_{wl}_{k}: vgpr={g.n_vgpr} lds={g.lds_bytes}")
