#!/usr/bin/env python3
"""bench.py -- headline benchmark: 2048-bit Paillier decryptions/s on N MI355X (BASELINE.json `metric`), plus the other
four BASELINE configs as `extra_configs` of the same JSON line.

One "step" = one pass of the hot path (SecretKey.Decrypt, paillier.go:292-303, for a whole batch) over one batch of
synthetic ciphertexts that are already resident in HBM (big-endian, element-major: the C-ABI operand format).  The timed
region contains everything a caller pays per batch: unpack -> CRT modexp over p^2 and q^2 (the VM kernel) -> L / CRT
recombination -> pack.  Results are checked bit-exactly after the timed region (decrypt(encrypt(m, r)) == m for the full
batch on every rank, plus sample checks on rank 0 in the CPU leg).

N > 1: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one process per GPU, RCCL) unless it is
already running under a launcher (WORLD_SIZE set).  Ciphertext batches shard with no data-path collective (scaling = weak:
every rank decrypts its own `--batch`); RCCL carries the barrier, the MAX-over-ranks time and -- in the threshold extra
config -- the one real exchange of the domain: the all-gather of the partial decryptions before the local combine.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measured on MI355X: profiles/r01_valu_rates.txt (v_mad_u64_u32, 4 waves/SIMD): 36.0 T lane-ops/s.
# Theoretical issue limit: 256 CU x 4 SIMD x 64 lanes / 4 cycles x 2.4 GHz = 39.3 T/s.  The roofline uses the
# theoretical figure (the stricter denominator).
PEAK_MAD_PER_S = 256 * 4 * 64 / 4 * 2.4e9
HBM_PEAK_GBPS = 8000.0


def alg_mul32_per_modexp(mod_bits: int, exp_bits: int, w: int = 5) -> float:
    """SURVEY.md §8(d) unit: CIOS on W 32-bit words = 2W^2+W multiply-adds; fixed window w:
    e + ceil(e/w) + 2^w - 2 + 2 Montgomery products."""
    W = mod_bits // 32
    return (exp_bits + -(-exp_bits // w) + (1 << w)) * (2 * W * W + W)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD process (torch.distributed.run, one rank per
    GPU) before this process has made any GPU call, and hand back its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def measure_traffic_pmc(bits: int, batch: int, timeout_s: float = 150.0):
    """HBM bytes per launch of the dominant kernel, measured for THIS run's workload: two child processes of this same
    command under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need a pass each: MI355X_MICROARCH.md "rocprofv3 PMC slots";
    no tracing domain is combined with --pmc), started before this process touches the GPU.  gfx950 correction of that
    guide: FETCH_SIZE counts half the bytes of wide coalesced reads -> 2 x FETCH_SIZE + WRITE_SIZE (both in KB).
    Returns (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, "already running under a profiler"
    out = {}
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "1", "--bits", str(bits), "--batch",
                   str(batch), "--headline-only"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            kern = None
            for line in r.stdout.decode(errors="ignore").splitlines():
                if line.startswith("{"):
                    kern = json.loads(line)["roofline"]["kernel_name"]
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kern and row["Kernel_Name"].startswith(kern) and row["Counter_Name"] == counter:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, f"no {counter} rows for kernel {kern}"
            vals = sorted(per.values())
            out[counter] = vals[len(vals) // 2]          # median over the launches of the dominant kernel (KB)
        return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, (
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one child pass each of this command; 2 x FETCH_SIZE + WRITE_SIZE "
            "(gfx950 half-count correction), median per launch of the dominant kernel")
    except Exception as e:  # noqa: BLE001 -- the measurement is optional; the bench line is not
        return None, f"PMC pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_clock_pmc(bits: int, batch: int, timeout_s: float = 150.0):
    """Sustained clock of the dominant kernel's launches in a child pass of this command under `rocprofv3 --pmc
    GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES` (no tracing domain): effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration
    (MI355X_MICROARCH.md "DVFS give-back"; within 3 % of the in-kernel clock on dispatches of 10 ms or more), and how much of the
    launch its waves were resident: SQ_WAVE_CYCLES / (SQ_WAVES x kernel cycles) -- 0.98 when the co-resident waves of a SIMD
    finish together, 0.75 when one of them runs alone for the second half.  A slow bench run is thereby
    attributable: low clock_ghz = the box / power management, low wave_cycles_ratio = scheduling.  Note the profiled pass itself
    clocks a few per cent below an un-profiled run (same guide).  Returns a dict or None."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_clk_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAVES", "--output-format", "csv", "-d", tmp, "-o", "p",
               "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--bits", str(bits),
               "--batch", str(batch), "--headline-only"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=timeout_s)
        if r.returncode != 0:
            return None
        kern = None
        for line in r.stdout.decode(errors="ignore").splitlines():
            if line.startswith("{"):
                kern = json.loads(line)["roofline"]["kernel_name"]
        per = {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern and row["Kernel_Name"].startswith(kern):
                    d = per.setdefault(row["Dispatch_Id"], {"s": None, "e": None})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if "Start_Timestamp" in row and "End_Timestamp" in row:
                        d["s"], d["e"] = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        clocks, ratios = [], []
        for d in per.values():
            if d["s"] is None or d["e"] is None or d["e"] - d["s"] < 5e6 or "GRBM_GUI_ACTIVE" not in d:
                continue                                  # (the quotient reads high on short dispatches: the table build etc.)
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            clocks.append(cyc / (d["e"] - d["s"]))
            if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES"):
                ratios.append(d["SQ_WAVE_CYCLES"] / (d["SQ_WAVES"] * cyc))
        if not clocks:
            return None
        clocks.sort()
        ratios.sort()
        return {"clock_ghz": clocks[len(clocks) // 2], "launches": len(clocks),
                "wave_cycles_ratio_raw": ratios[len(ratios) // 2] if ratios else None,
                # the SQ counters of this pass cover one shader engine in four (raw 0.245 = 0.98 / 4 on the headline launch)
                "wave_cycles_ratio": 4.0 * ratios[len(ratios) // 2] if ratios else None,
                "source": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES, one child pass of this command (headline only); "
                          "GRBM_GUI_ACTIVE / 8 / dispatch duration, median over the dominant kernel's launches; the ratio is "
                          "SQ_WAVE_CYCLES / (SQ_WAVES x kernel cycles) as the counters come (SQ counters are sampled per shader "
                          "engine: compare runs with each other, not with 1.0)"}
    except Exception:  # noqa: BLE001 -- optional
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_cpu_info():
    """CPU model / sockets / cores / threads of the host and what this process may use of it."""
    info = {"model": None, "sockets": None, "physical_cores": None, "logical_cpus": os.cpu_count()}
    try:
        phys, socks = set(), set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and info["model"] is None:
                info["model"] = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
                socks.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
                phys.add((pid, cid))
        info["sockets"], info["physical_cores"] = len(socks) or None, len(phys) or None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0))
    try:   # cgroup v2 CPU quota of the box's share
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            info["cgroup_cpu_quota"] = float(q) / float(per)
            usable = min(usable, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    info["usable_cpus"] = usable
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="ciphertexts per GPU per step")
    ap.add_argument("--bits", type=int, default=2048, choices=[1024, 2048, 3072])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (wall seconds)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs (BASELINE configs 2-5)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--headline-only", action="store_true", help="(internal: PMC child pass) headline only, no CPU leg")
    ap.add_argument("--extra-steps", type=int, default=2)
    ap.add_argument("--deep-queue", action="store_true",
                    help="also time the headline path on 131072 ciphertexts (extra config decrypt_2048_b131072); off by default so "
                         "that the kernel-trace average of vm_asm_37_16 in the default run is the headline launch")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 and exchange over gloo: rehearses the multi-rank code path (self-launch, "
                         "sharding, the threshold exchange, the reductions) on a one-GPU box; not a measurement")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_extra = args.no_traffic = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))          # nothing has touched the GPU yet

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)

    traffic, traffic_note, clock = None, "not measured (--no-traffic / N > 1)", None
    if world == 1 and not args.no_traffic:
        traffic, traffic_note = measure_traffic_pmc(args.bits, args.batch)   # child processes; before any GPU call here
        clock = measure_clock_pmc(args.bits, args.batch)

    import numpy as np
    import torch

    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            print(f"[bench] process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

    import paillier_amd as pa
    from paillier_amd.api import MEM_DEVICE
    from paillier_amd import dist as pdist

    with open(os.path.join(ROOT, "tests", "golden", "keys.json")) as f:
        KEYS = json.load(f)

    # the committed plan table: executed multiply-adds per unit of every config at its default shape on one GPU.  A planning
    # predicate (paillier_amd/csrc/plan.hpp) that moves a shape onto another ladder shows up here before it shows up in a timing.
    try:
        with open(os.path.join(ROOT, "profiles", "plan_table.json")) as f:
            PLAN = json.load(f)
    except OSError:
        PLAN = {"per_unit": {}, "tolerance": 0.01}
    plan_dev = {}
    RANK_FLOOR = {}
    for rf in ("r05_rank_floor.json", "r04_rank_floor.json"):      # the newest table tools/rank_floor.py wrote
        try:
            with open(os.path.join(ROOT, "profiles", rf)) as f:
                RANK_FLOOR = json.load(f)
            break
        except OSError:
            continue

    def plan_check(name, mads_per_unit, applies=True):
        """fraction by which this run's executed multiply-adds per unit differ from the committed table (None: not comparable)"""
        want = PLAN["per_unit"].get(name)
        if not applies or not want or not mads_per_unit:
            return None
        d = mads_per_unit / want - 1.0
        if abs(d) > PLAN.get("tolerance", 0.01):
            plan_dev[name] = {"executed_mad28_per_unit": mads_per_unit, "expected": want, "deviation": d}
        return d

    def paillier_key(bits):
        k = KEYS["paillier"][str(bits)]
        p, q = int(k["p"], 16), int(k["q"], 16)
        return p, q, p * q, (p - 1) * (q - 1)

    p, q, n, lam = paillier_key(args.bits)
    ctx = pa.Context(local_rank, torch.cuda.current_stream().cuda_stream)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    assert sk.has_crt
    B = args.batch
    pb, cb = pk.plain_bytes(), pk.cipher_bytes()

    def rand_below(modulus, count, nbytes, rng):
        """uniform-ish values below `modulus` as big-endian rows: top byte strictly below the modulus' top byte"""
        raw = rng.integers(0, 256, size=(count, nbytes), dtype=np.uint8)
        top = modulus >> (8 * (nbytes - 1))
        raw[:, 0] %= np.uint8(min(max(top, 1), 255))
        return raw

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- headline: Decrypt ---------------------------------------------------------------------------------------
    # synthetic inputs: uniform m in [0, n) and r in Z_n^* from a seeded PRNG (seed differs per rank);
    # ciphertexts are produced by the engine's own Encrypt (setup, untimed).
    rng = np.random.default_rng(1234 + rank)
    m_host = rand_below(n, B, pb, rng)
    r_host = rand_below(n, B, pb, rng)
    r_host[:, -1] |= 1
    m_dev = torch.from_numpy(m_host).to(dev)
    r_dev = torch.from_numpy(r_host).to(dev)
    c_dev = torch.zeros((B, cb), dtype=torch.uint8, device=dev)
    out_dev = torch.zeros((B, pb), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, m_dev.data_ptr(), pb, r_dev.data_ptr(), pb, c_dev.data_ptr(), cb, MEM_DEVICE)
    enc_prof = ctx.last_profile()

    def step():
        sk.decrypt_raw(B, c_dev.data_ptr(), cb, out_dev.data_ptr(), pb, MEM_DEVICE)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    vm_ms = []
    for _ in range(args.steps):
        step()
        vm_ms.append(ctx.last_profile()["vm_ms"])  # HIP events on the launch stream; already complete (blocking API)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    elapsed = pdist.max_over_ranks(elapsed, dev)
    prof = ctx.last_profile()

    # parity at full size: round trip is the identity on every lane, on every rank
    ok = bool(torch.equal(out_dev, m_dev))
    if not ok:
        bad = int((out_dev != m_dev).any(dim=1).sum().item())
        raise SystemExit(f"[bench] rank {rank}: PARITY FAILURE: {bad} of {B} decryptions differ from the plaintexts")

    # ---- extra configs (BASELINE.json configs 2-5), same process, same rules: inputs resident in HBM, HIP-event kernel
    # time, executed multiply-adds from the library's per-opcode count, full-batch round-trip parity where one exists ------
    extras, checks = [], {}

    def timed(fn, steps):
        """(seconds per call -- MAX over ranks, between barriers --, this rank's VM-kernel ms, executed multiply-adds, kernel)"""
        fn()
        fn()    # (a call that grew the library's workspace is followed by one that may consolidate it: both are warm-up)
        barrier()
        ms, mads, kern = [], 0.0, ""
        t = time.perf_counter()
        for _ in range(steps):
            mads, vms, kern = fn()
            ms.append(vms)
        barrier()
        return pdist.max_over_ranks(time.perf_counter() - t, dev) / steps, sum(ms) / len(ms), mads, kern

    def one_call(fn):
        """run one C-ABI call and return (mads, vm_ms, kernel) of it"""
        fn()
        pr_ = ctx.last_profile()
        return pr_["vm_mads"], pr_["vm_ms"], pr_["kernel"]

    def entry(name, workload, unit, count, dt, vms, mads, kern, parity, scaling="weak"):
        """count = units of the WHOLE job (all ranks) per call; dt = MAX-over-ranks seconds per call; vms / mads = rank 0's
        kernel time and executed multiply-adds (its `frac` is therefore a per-GPU figure on every N)"""
        e = {"config": name, "workload": workload, "value": count / dt, "unit": unit, "ms_per_batch": dt * 1e3,
             "kernel": kern, "kernel_ms_per_batch": vms, "executed_mad28": mads,
             "frac": (mads / (vms * 1e-3) / PEAK_MAD_PER_S) if vms else None, "parity": parity}
        # the same multiplies over the CALL's wall time (rank 0's share of it): `frac` divides by the SUM of the VM launches' own
        # durations -- a call whose launches overlap (the DDLEQ prover runs a side-stream ladder beside its first one) sums the
        # overlapped time twice there, and a call with time between its ladders hides that time there
        if vms and world == 1:
            e["frac_of_call_time"] = mads / dt / PEAK_MAD_PER_S
        if world > 1:
            e.update({"scaling": scaling, "n_gpus": world})
            # a strong-scaled entry's rank runs a SMALL shard: what that shard takes on one GPU (measured, profiles/r0N_rank_floor.json)
            fl = RANK_FLOOR.get(name, {}).get(str(world))
            if scaling == "strong" and fl:
                e["predicted_rank_floor_ms"] = fl
                e["predicted_rank_floor_note"] = RANK_FLOOR[name].get("unit", "")
        # rank 0's executed multiply-adds per unit of ITS share against the committed plan table (weak-scaled configs keep their
        # per-rank shape on every N; a strong-scaled config's per-rank shape -- and plan -- changes with N: checked at N = 1 only)
        units_rank0 = count / world
        e["executed_mad28_per_unit"] = mads / units_rank0 if units_rank0 else None
        e["plan_deviation"] = plan_check(name, e["executed_mad28_per_unit"], applies=(world == 1 or scaling == "weak"))
        # a config whose algorithm needs fewer multiplies than the ladders of the reference's formula (the prover through the
        # structure of the unit group): its `frac` prices only what it executes; the rate in units of the LITERAL algorithm's
        # multiplies says what the saving is worth (above the executed fraction: that much work is simply not done)
        lit = PLAN.get("literal_per_unit", {}).get(name)
        if lit and world == 1:
            e["literal_algorithm"] = {"mad28_per_unit": lit, "equivalent_frac_of_call_time": lit * count / dt / PEAK_MAD_PER_S,
                                      "multiplies_saved": 1.0 - e["executed_mad28_per_unit"] / lit}
        return e

    def all_ranks_ok(flag, what):
        if not pdist.min_over_ranks(1 if flag else 0, dev):
            raise SystemExit(f"[bench] {what}" + (" (on some rank)" if world > 1 else ""))

    if not args.no_extra:
        ES = args.extra_steps
        # config 2: Batch 65536 Encrypt, 2048-bit -- per GPU (weak scaling: independent ciphertexts, no exchange)
        p2, q2, n2k, lam2 = paillier_key(2048)
        pk2 = pk if args.bits == 2048 else pa.PublicKey(ctx, n2k, n2k + 1)
        sk2 = sk if args.bits == 2048 else pa.SecretKey(ctx, pk2, lam2)
        BE = 65536
        rg = np.random.default_rng(2 + 1000 * rank)
        em_h, er_h = rand_below(n2k, BE, 256, rg), rand_below(n2k, BE, 256, rg)
        er_h[:, -1] |= 1
        em, er = torch.from_numpy(em_h).to(dev), torch.from_numpy(er_h).to(dev)
        ec = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
        eo = torch.zeros((BE, 256), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.encrypt_with_r_raw(
            BE, em.data_ptr(), 256, er.data_ptr(), 256, ec.data_ptr(), 512, MEM_DEVICE)), ES)
        sk2.decrypt_raw(BE, ec.data_ptr(), 512, eo.data_ptr(), 256, MEM_DEVICE)
        all_ranks_ok(torch.equal(eo, em), "Encrypt-2048: Decrypt(Encrypt(m, r)) != m")
        extras.append(entry("encrypt_2048", "Batch 65536 EncryptWithR per GPU, 2048-bit n, level 1 (r^n * (1+n)^m mod n^2)",
                            "encryptions/s", world * BE, dt, vms, mads, kern, "65536-lane decrypt round trip on every rank"))
        if world == 1:
            # the same batch with the library drawing r itself (PublicKey.Encrypt, paillier.go:258-269): getrandom(2) on host
            # threads + rejection below n, gcd test on the device
            ec2 = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
            for _ in range(2):      # warm-up as in timed(): the first calls grow the library's workspace and start its host threads
                pk2.encrypt_raw(BE, em.data_ptr(), 256, ec2.data_ptr(), 512, None, 0, MEM_DEVICE)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(ES):
                pk2.encrypt_raw(BE, em.data_ptr(), 256, ec2.data_ptr(), 512, None, 0, MEM_DEVICE)
            torch.cuda.synchronize()
            dt_rng = (time.perf_counter() - t) / ES
            sk2.decrypt_raw(BE, ec2.data_ptr(), 512, eo.data_ptr(), 256, MEM_DEVICE)
            assert torch.equal(eo, em), "[bench] Encrypt (library randomness): Decrypt(Encrypt(m)) != m"
            extras[-1].update({"encrypt_with_library_randomness_per_s": BE / dt_rng, "fraction_of_with_r": dt / dt_rng})
            checks["encrypt_2048"] = (n2k, em_h[:64], er_h[:64], ec[:64].cpu().numpy())
            # the same call on the SECRET key (SecretKey embeds PublicKey in the reference): r^n through p^2 and q^2
            dt_sk, vms_sk, mads_sk, kern_sk = timed(lambda: one_call(lambda: sk2.encrypt_with_r_raw(
                BE, em.data_ptr(), 256, er.data_ptr(), 256, ec2.data_ptr(), 512, MEM_DEVICE)), ES)
            assert torch.equal(ec2, ec), "[bench] key holder's EncryptWithR differs from the public path"
            extras[-1].update({"key_holder_encryptions_per_s": BE / dt_sk, "key_holder_kernel": kern_sk})
            del ec2

        # the headline kernel with a deeper queue: 131072 ciphertexts = four waves' worth of work per SIMD slot pair, so the
        # hardware dispatcher refills CUs as their first workgroups retire (at 65536 every workgroup is resident from the
        # start and the launch lasts as long as its slowest CU).  Same Decrypt, same key; the ciphertexts are the Encrypt
        # batch above twice.
        if args.deep_queue and world == 1:
            BB = 2 * BE
            bc = torch.cat([ec, ec]).contiguous()
            bo = torch.zeros((BB, 256), dtype=torch.uint8, device=dev)
            dt, vms, mads, kern = timed(lambda: one_call(lambda: sk2.decrypt_raw(BB, bc.data_ptr(), 512, bo.data_ptr(), 256, MEM_DEVICE)), ES)
            assert torch.equal(bo[:BE], em) and torch.equal(bo[BE:], em), "[bench] Decrypt-2048 x 131072 round trip failed"
            extras.append(entry("decrypt_2048_b131072", "Batch 131072 Decrypt, 2048-bit n, level 1, CRT (the headline path, twice the batch)",
                                "decryptions/s", BB, dt, vms, mads, kern, "131072-lane round trip"))
            del bc, bo
        # ---- the homomorphic operations (operations.go:11-64; the reference's own BenchmarkAdd / BenchmarkConstMul2,
        # operations_test.go:165-198) on the ciphertexts of config 2: 65536 per GPU, weak-scaled like Encrypt ----------------------
        hb = torch.flip(ec, dims=[0]).contiguous()                   # second operand: the same ciphertexts in reverse order
        ho = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
        ho2 = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)

        def hbm_note(e_, bytes_per_unit):
            # Add / Sub are a handful of multiplies per 1.5 KB moved: the HBM side of the roofline is the one to read
            e_["alg_bytes_per_unit"] = bytes_per_unit
            e_["hbm_frac_of_call_time"] = bytes_per_unit * BE / (e_["ms_per_batch"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            return e_

        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.add_raw(BE, ec.data_ptr(), 512, hb.data_ptr(), 512, ho.data_ptr(), 512, MEM_DEVICE)), ES)
        add_rows = ho[:64].cpu().numpy()
        extras.append(hbm_note(entry("add_2048", "65536 Add(a, b) per GPU, 2048-bit n, level 1: a * b mod n^2 (operations.go:11-29; "
                                     "BenchmarkAdd, operations_test.go:165-172)", "additions/s", world * BE, dt, vms, mads, kern,
                                     "Sub(Add(a, b), b) == a on all 65536 lanes (below)"), 3 * 512))
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.sub_raw(BE, ho.data_ptr(), 512, hb.data_ptr(), 512, ho2.data_ptr(), 512, MEM_DEVICE)), ES)
        all_ranks_ok(torch.equal(ho2, ec), "Sub(Add(a, b), b) != a")
        sub_rows = ho2[:64].cpu().numpy()
        extras.append(hbm_note(entry("sub_2048", "65536 Sub(a, b) per GPU, 2048-bit n, level 1: a * b^-1 mod n^2 (operations.go:32-55): "
                                     "batch inversion as a product tree, one inversion of the root on the host", "subtractions/s",
                                     world * BE, dt, vms, mads, kern, "Sub(Add(a, b), b) == a on all 65536 lanes"), 3 * 512))
        # ConstMult with the reference benchmark's constant (BenchmarkConstMul2: k = 50^50 mod n^2 -- 283 bits, shared by the batch)
        k50 = pow(50, 50, n2k * n2k)
        k50b = np.frombuffer(k50.to_bytes((k50.bit_length() + 7) // 8, "big"), dtype=np.uint8).copy()
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.const_mult_raw(BE, ec.data_ptr(), 512, k50b, k50b.size, 0, ho.data_ptr(), 512, MEM_DEVICE)), ES)
        # ... the same constant as one exponent PER ciphertext (another ladder: per-number windows): the same ciphertexts out
        k50rows = torch.from_numpy(np.tile(k50b, (BE, 1))).to(dev)
        pk2.const_mult_raw(BE, ec.data_ptr(), 512, k50rows.data_ptr(), k50b.size, k50b.size, ho2.data_ptr(), 512, MEM_DEVICE)
        all_ranks_ok(torch.equal(ho, ho2), "ConstMult: shared k and per-ciphertext k disagree")
        extras.append(entry("const_mult_2048", "65536 ConstMult(c, k) per GPU, 2048-bit n, k = 50^50 mod n^2 shared (283 bits; "
                            "BenchmarkConstMul2, operations_test.go:186-198): c^k mod n^2 (operations.go:58-64)", "ciphertexts/s",
                            world * BE, dt, vms, mads, kern, "all 65536 == the per-ciphertext-exponent ladder on the same k"))
        if world == 1:
            checks["homomorphic_2048"] = (n2k, k50, ec[:64].cpu().numpy(), hb[:64].cpu().numpy(), ho[:64].cpu().numpy(), add_rows, sub_rows)
        # ... and a full-width k per ciphertext (k < n^2: what NestedAdd / a plaintext-sized scalar costs)
        kf_h = rand_below(n2k * n2k, BE, 512, np.random.default_rng(21 + 1000 * rank))
        kf = torch.from_numpy(kf_h).to(dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.const_mult_raw(BE, ec.data_ptr(), 512, kf.data_ptr(), 512, 512, ho.data_ptr(), 512, MEM_DEVICE)), ES)
        # parity at full size: Decrypt(c^k) == k * m mod n  <=>  Decrypt(c^k) == Decrypt(c^(k mod n)): checked through the group law
        # c^k * c^(k') == c^(k + k') on a second exponent k' = 2^4096 - 1 - k (the bytewise complement): the product is c^(2^4096 - 1)
        kc = (255 - kf).contiguous()
        pk2.const_mult_raw(BE, ec.data_ptr(), 512, kc.data_ptr(), 512, 512, ho2.data_ptr(), 512, MEM_DEVICE)
        hx = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
        pk2.add_raw(BE, ho.data_ptr(), 512, ho2.data_ptr(), 512, hx.data_ptr(), 512, MEM_DEVICE)
        ones = np.full(512, 255, dtype=np.uint8)
        hw = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
        pk2.const_mult_raw(BE, ec.data_ptr(), 512, ones, 512, 0, hw.data_ptr(), 512, MEM_DEVICE)
        all_ranks_ok(torch.equal(hx, hw), "ConstMult: c^k * c^(2^4096 - 1 - k) != c^(2^4096 - 1)")
        del hx
        extras.append(entry("const_mult_2048_full_k", "65536 ConstMult(c, k) per GPU, 2048-bit n, one full-width k < n^2 per ciphertext "
                            "(4096-bit exponents, per-number windows)", "ciphertexts/s", world * BE, dt, vms, mads, kern,
                            "c^k * c^(2^4096 - 1 - k) == c^(2^4096 - 1) on all 65536 lanes"))
        if world == 1:
            checks["const_mult_full_2048"] = (n2k, ec[:32].cpu().numpy(), kf_h[:32], ho[:32].cpu().numpy())
        del hb, ho, ho2, hw, kf, kc, k50rows

        # AltEncryptWithRAtLevel (paillier.go:221-238): c = G^m * h^(r mod K) mod n^2 with h = (N - H)^N fixed per key -- a fixed-base
        # comb table, no squarings.  65536 per GPU.
        k2 = KEYS["paillier"]["2048"]
        pka = pa.PublicKey(ctx, n2k, n2k + 1, int(k2["h"], 16), int(k2["k"], 16))
        ac = torch.zeros((BE, 512), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pka.alt_encrypt_with_r_raw(
            BE, em.data_ptr(), 256, er.data_ptr(), 256, ac.data_ptr(), 512, None, MEM_DEVICE)), ES)
        sk2.decrypt_raw(BE, ac.data_ptr(), 512, eo.data_ptr(), 256, MEM_DEVICE)
        all_ranks_ok(torch.equal(eo, em), "AltEncrypt: Decrypt(AltEncrypt(m, r)) != m")
        extras.append(entry("alt_encrypt_2048", "65536 AltEncryptWithR per GPU, 2048-bit n, level 1: (1+n)^m * h^(r mod K) mod n^2 "
                            "(paillier.go:221-238)", "encryptions/s", world * BE, dt, vms, mads, kern,
                            "65536-lane decrypt round trip on every rank"))
        if world == 1:
            checks["alt_encrypt_2048"] = (n2k, int(k2["h"], 16), int(k2["k"], 16), em_h[:64], er_h[:64], ac[:64].cpu().numpy())
        del ac, pka
        del em, er, ec, eo

        # config 3: Batch 65536 Decrypt, 3072-bit -- per GPU (weak scaling)
        p3, q3, n3k, lam3 = paillier_key(3072)
        pk3 = pa.PublicKey(ctx, n3k, n3k + 1)
        sk3 = pa.SecretKey(ctx, pk3, lam3)
        rg = np.random.default_rng(3 + 1000 * rank)
        dm_h, dr_h = rand_below(n3k, BE, 384, rg), rand_below(n3k, BE, 384, rg)
        dr_h[:, -1] |= 1
        dm, dr = torch.from_numpy(dm_h).to(dev), torch.from_numpy(dr_h).to(dev)
        dc = torch.zeros((BE, 768), dtype=torch.uint8, device=dev)
        do = torch.zeros((BE, 384), dtype=torch.uint8, device=dev)
        pk3.encrypt_with_r_raw(BE, dm.data_ptr(), 384, dr.data_ptr(), 384, dc.data_ptr(), 768, MEM_DEVICE)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: sk3.decrypt_raw(
            BE, dc.data_ptr(), 768, do.data_ptr(), 384, MEM_DEVICE)), ES)
        all_ranks_ok(torch.equal(do, dm), "Decrypt-3072 round trip failed")
        extras.append(entry("decrypt_3072", "Batch 65536 Decrypt per GPU, 3072-bit n, level 1, CRT over p^2, q^2", "decryptions/s",
                            world * BE, dt, vms, mads, kern, "65536-lane round trip on every rank"))
        if world == 1:
            checks["decrypt_3072"] = (n3k, lam3, dc[:32].cpu().numpy(), do[:32].cpu().numpy())
        del pk3, sk3, dm, dr, dc, do

        # level-two (Damgard-Jurik s = 2) EncryptWithR, 16384 ciphertexts per GPU: r^(n^2) (1+n)^m mod n^3 on the three-digit kernel
        BL = 16384
        rg = np.random.default_rng(6 + 1000 * rank)
        lm_h, lr_h = rand_below(n2k * n2k, BL, 512, rg), rand_below(n2k, BL, 256, rg)
        lr_h[:, -1] |= 1
        lm, lr = torch.from_numpy(lm_h).to(dev), torch.from_numpy(lr_h).to(dev)
        lc = torch.zeros((BL, 768), dtype=torch.uint8, device=dev)
        lo = torch.zeros((BL, 512), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.encrypt_with_r_raw(
            BL, lm.data_ptr(), 512, lr.data_ptr(), 256, lc.data_ptr(), 768, MEM_DEVICE, level=1)), ES)
        sk2.decrypt_raw(BL, lc.data_ptr(), 768, lo.data_ptr(), 512, MEM_DEVICE, level=1)
        all_ranks_ok(torch.equal(lo, lm), "level-two Encrypt: Decrypt(Encrypt(m, r)) != m")
        extras.append(entry("encrypt_l2_2048", "Batch 16384 level-two EncryptWithR per GPU, 2048-bit n (r^(n^2) * (1+n)^m mod n^3)",
                            "encryptions/s", world * BL, dt, vms, mads, kern, "16384-lane level-two decrypt round trip on every rank"))
        if world == 1:
            checks["encrypt_l2_2048"] = (n2k, lm_h[:32], lr_h[:32], lc[:32].cpu().numpy())
            # the same call on the SECRET key (SecretKey embeds PublicKey): r^(n^2) mod n^3 as the Teichmueller lift of
            # (r mod p)^(q^2 mod (p - 1)) (and likewise modulo q^3) -- the same ciphertexts, a fifth of the multiplies
            lc2 = torch.zeros((BL, 768), dtype=torch.uint8, device=dev)
            dt_sk, vms_sk, mads_sk, kern_sk = timed(lambda: one_call(lambda: sk2.encrypt_with_r_raw(
                BL, lm.data_ptr(), 512, lr.data_ptr(), 256, lc2.data_ptr(), 768, MEM_DEVICE, level=1)), ES)
            assert torch.equal(lc2, lc), "[bench] key holder's level-two EncryptWithR differs from the public path"
            extras[-1].update({"key_holder_encryptions_per_s": BL / dt_sk, "key_holder_kernel": kern_sk,
                               "key_holder_executed_mad28_per_unit": mads_sk / BL})
            del lc2
        # level-two Decrypt (paillier.go:292-340 with s = 2: what NestedDecrypt runs first, :344-372) of the ciphertexts just made:
        # CRT over p^3, q^3 on the three-digit kernel, the recovery algorithm's integers
        dt, vms, mads, kern = timed(lambda: one_call(lambda: sk2.decrypt_raw(BL, lc.data_ptr(), 768, lo.data_ptr(), 512, MEM_DEVICE, level=1)), ES)
        all_ranks_ok(torch.equal(lo, lm), "level-two Decrypt: Decrypt(Encrypt(m, r)) != m")
        extras.append(entry("decrypt_l2_2048", "Batch 16384 level-two Decrypt per GPU, 2048-bit n (c^(p-1) mod p^3, c^(q-1) mod q^3, the "
                            "Damgard-Jurik recovery, Garner)", "decryptions/s", world * BL, dt, vms, mads, kern,
                            "16384-lane round trip on every rank"))
        if world == 1:
            checks["decrypt_l2_2048"] = (n2k, lam2, lc[:32].cpu().numpy(), lo[:32].cpu().numpy())
        # NestedAdd (operations.go:121-127): a level-two ciphertext raised to a level-one ciphertext VALUE = ConstMult at level two
        # with one 4096-bit k per ciphertext
        nk_h = rand_below(n2k * n2k, BL, 512, np.random.default_rng(61 + 1000 * rank))
        nk = torch.from_numpy(nk_h).to(dev)
        no = torch.zeros((BL, 768), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.const_mult_raw(BL, lc.data_ptr(), 768, nk.data_ptr(), 512, 512, no.data_ptr(), 768,
                                                                                 MEM_DEVICE, level=1)), ES)
        # parity at full size: Decrypt_2(ct^k) = k * m mod n^2 -- checked through the group law as for ConstMult: ct^k * ct^(2^4096-1-k) == ct^(2^4096-1)
        nkc = (255 - nk).contiguous()
        no2 = torch.zeros((BL, 768), dtype=torch.uint8, device=dev)
        pk2.const_mult_raw(BL, lc.data_ptr(), 768, nkc.data_ptr(), 512, 512, no2.data_ptr(), 768, MEM_DEVICE, level=1)
        no3 = torch.zeros((BL, 768), dtype=torch.uint8, device=dev)
        pk2.add_raw(BL, no.data_ptr(), 768, no2.data_ptr(), 768, no3.data_ptr(), 768, MEM_DEVICE, level=1)
        pk2.const_mult_raw(BL, lc.data_ptr(), 768, np.full(512, 255, dtype=np.uint8), 512, 0, no2.data_ptr(), 768, MEM_DEVICE, level=1)
        all_ranks_ok(torch.equal(no3, no2), "NestedAdd: ct^k * ct^(2^4096 - 1 - k) != ct^(2^4096 - 1)")
        extras.append(entry("nested_add_2048", "Batch 16384 NestedAdd per GPU (operations.go:121-127): level-two ciphertext ^ (4096-bit value of a "
                            "level-one ciphertext) mod n^3 = ConstMult at level two, one k per ciphertext", "ciphertexts/s", world * BL, dt, vms,
                            mads, kern, "ct^k * ct^(2^4096 - 1 - k) == ct^(2^4096 - 1) on all 16384 lanes"))
        if world == 1:
            checks["nested_add_2048"] = (n2k, lc[:16].cpu().numpy(), nk_h[:16], no[:16].cpu().numpy())
        del nk, nkc, no, no2, no3
        del lm, lr, lc, lo

        # config 5: DDLEQ prove / verify, 2048-bit, 16384 (statement, instance) pairs IN TOTAL -- strong scaling: the pairs are
        # split over the ranks (shard_slice), every rank proves and verifies its own slice, no exchange (independent proofs)
        BD_ALL = 16384
        cb3, pb2 = pk2.cipher_bytes(1), pk2.plain_bytes(1)
        tb = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(dev)

        def ddleq_statements(count, seed):
            """`count` statements (ct1, ct2 = NestedRandomize(ct1; a, b), a, b) of the WHOLE job from one seed; returns this
            rank's slice (device tensors) and the NestedRandomize timing of it"""
            rg_ = np.random.default_rng(seed)
            def unit(k):
                a_ = rand_below(n2k, k, 256, rg_)
                a_[:, -1] |= 1
                return a_
            msg, r1, r2, a_h, b_h = rand_below(n2k, count, 256, rg_), unit(count), unit(count), unit(count), unit(count)
            lo_, hi_ = pdist.shard_slice(count, rank, world)
            k = hi_ - lo_
            inner = torch.zeros((k, 512), dtype=torch.uint8, device=dev)
            c1 = torch.zeros((k, cb3), dtype=torch.uint8, device=dev)
            pk2.encrypt_with_r_raw(k, tb(msg[lo_:hi_]).data_ptr(), 256, tb(r1[lo_:hi_]).data_ptr(), 256, inner.data_ptr(), 512, MEM_DEVICE)
            pk2.encrypt_with_r_raw(k, inner.data_ptr(), 512, tb(r2[lo_:hi_]).data_ptr(), 256, c1.data_ptr(), cb3, MEM_DEVICE, level=1)
            a_d, b_d = tb(a_h[lo_:hi_]), tb(b_h[lo_:hi_])
            c2 = torch.zeros((k, cb3), dtype=torch.uint8, device=dev)
            # ct2 = NestedRandomize(ct1; a, b) = ct1^(a^n mod n^2) * b^(n^2) mod n^3  (operations.go:96-118)
            tm_ = timed(lambda: one_call(lambda: pk2.nested_randomize_with_ab_raw(
                k, c1.data_ptr(), a_d.data_ptr(), b_d.data_ptr(), c2.data_ptr(), MEM_DEVICE)), 1)
            return k, c1, c2, a_d, b_d, (lambda kk: unit(kk)), tm_

        BD, ct1, ct2, da, db, unit, (dt, vms, mads, kern) = ddleq_statements(BD_ALL, 5)
        extras.append(entry("nested_randomize_2048", "16384 NestedRandomize (a, b supplied), 2048-bit n: ct^(a^n) * b^(n^2) mod n^3 "
                            "(operations.go:96-118), the statements of the DDLEQ config", "ciphertexts/s", BD_ALL, dt, vms, mads, kern,
                            "the DDLEQ prover's sanity check recomputes every ct2 (below)", scaling="strong"))
        lo_d, hi_d = pdist.shard_slice(BD_ALL, rank, world)
        dxy = unit(2 * BD_ALL)                                  # the draws x | y of the whole job; this rank's rows of each
        dx, dy = tb(dxy[:BD_ALL][lo_d:hi_d]), tb(dxy[BD_ALL:][lo_d:hi_d])
        al = torch.zeros((BD, cb3), dtype=torch.uint8, device=dev)
        pe = torch.zeros((BD, pb2), dtype=torch.uint8, device=dev)
        pf = torch.zeros((BD, cb3), dtype=torch.uint8, device=dev)
        okh = np.zeros(BD, dtype=np.int32)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: sk2.ddleq_prove_raw(
            BD, ct1.data_ptr(), ct2.data_ptr(), da.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(),
            pe.data_ptr(), pf.data_ptr(), MEM_DEVICE)), max(ES, 3))       # (a call's time varies by a few per cent: three of them)
        extras.append(entry("ddleq_prove_2048", "16384 DDLEQ instances (secpar = 1 each), 2048-bit n: sanity check, Alpha, "
                            "Fiat-Shamir bit, response through level-two ExtractRandonness (ddleq.go:55-127)",
                            "instances/s", BD_ALL, dt, vms, mads, kern, "every proof verifies (below)", scaling="strong"))
        dt, vms, mads, kern = timed(lambda: one_call(lambda: pk2.ddleq_verify_raw(
            BD, ct1.data_ptr(), ct2.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), okh,
            MEM_DEVICE)), ES)
        all_ranks_ok(bool(okh.all()), "DDLEQ: a proof made by the prover was rejected by the verifier")
        extras.append(entry("ddleq_verify_2048", "16384 DDLEQ instances, 2048-bit n: hash bit + check^(E^n) * F^(n^2) mod n^3 "
                            "(ddleq.go:129-153)", "instances/s", BD_ALL, dt, vms, mads, kern, "16384 of 16384 accepted", scaling="strong"))
        if world == 1:
            S = 8
            checks["ddleq_2048"] = (n2k, lam2, [x_[:S].cpu().numpy() for x_ in (ct1, ct2, da, db, dx, dy, al, pe, pf)])
            checks["nested_randomize_2048"] = (n2k, [x_[:32].cpu().numpy() for x_ in (ct1, da, db, ct2)])
        del ct1, ct2, da, db, dx, dy, al, pe, pf

        # the same config as the reference's own test drives it (ddleq_test.go:74-88): ProveDDLEQ with secpar = 40 -- 1536
        # statements x 40 instances = 61440 instances per call; what depends on the statement only is computed once per
        # statement (pgpu_ddleq_prove_secpar).  Statements split over the ranks.  (The size is chosen so that every launch of
        # the call fills whole waves per SIMD: a launch a few per cent over such a boundary runs up to twice as long --
        # DESIGN.md "wave quantisation" -- and hoisting only pays once the per-instance launches fill the chip.)
        SP, NS_ALL = 40, 1536
        NS, s1, s2, sa, sb, unit, _ = ddleq_statements(NS_ALL, 55)
        lo_s, hi_s = pdist.shard_slice(NS_ALL, rank, world)
        sxy = unit(2 * NS_ALL * SP)
        sx, sy = tb(sxy[:NS_ALL * SP][lo_s * SP:hi_s * SP]), tb(sxy[NS_ALL * SP:][lo_s * SP:hi_s * SP])
        BI = NS * SP
        al = torch.zeros((BI, cb3), dtype=torch.uint8, device=dev)
        pe = torch.zeros((BI, pb2), dtype=torch.uint8, device=dev)
        pf = torch.zeros((BI, cb3), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: sk2.ddleq_prove_secpar_raw(
            NS, SP, s1.data_ptr(), s2.data_ptr(), sa.data_ptr(), sb.data_ptr(), sx.data_ptr(), sy.data_ptr(), al.data_ptr(),
            pe.data_ptr(), pf.data_ptr(), MEM_DEVICE)), max(ES, 2))
        # verify: every instance against its statement (rows of ct1 / ct2 repeated per instance)
        rep = torch.arange(NS, device=dev).repeat_interleave(SP)
        okh = np.zeros(BI, dtype=np.int32)
        r1_, r2_ = s1[rep].contiguous(), s2[rep].contiguous()
        pk2.ddleq_verify_raw(BI, r1_.data_ptr(), r2_.data_ptr(), sx.data_ptr(), sy.data_ptr(), al.data_ptr(), pe.data_ptr(),
                             pf.data_ptr(), okh, MEM_DEVICE)
        all_ranks_ok(bool(okh.all()), "DDLEQ secpar 40: a proof made by the prover was rejected by the verifier")
        by_name = {e_["config"]: e_ for e_ in extras}
        e40 = entry("ddleq_prove_2048_secpar40", "1536 DDLEQ statements x secpar 40 = 61440 instances (ProveDDLEQ, ddleq.go:27-40, at the "
                    "reference's test setting ddleq_test.go:74-88), 2048-bit n: sanity check, a^n, a^-1, ExtractRandonness once per "
                    "statement; x^n, Alpha, Fiat-Shamir bit and the response per instance", "instances/s", NS_ALL * SP, dt, vms, mads,
                    kern, "all 61440 instances verify against their statements", scaling="strong")
        e40["speedup_over_secpar1_instances"] = e40["value"] / by_name["ddleq_prove_2048"]["value"]
        extras.append(e40)
        if world == 1:
            checks["ddleq_secpar40"] = (n2k, lam2, SP, [x_.cpu().numpy() for x_ in (s1[:1], s2[:1], sa[:1], sb[:1], sx[:SP], sy[:SP],
                                                                                    al[:SP], pe[:SP], pf[:SP])])
        del s1, s2, sa, sb, sx, sy, al, pe, pf, r1_, r2_

    # config 4: threshold decryption (t = 3, l = 5), 16384 ciphertexts per step over the `world` ranks -- strong scaling:
    # (server, ciphertext) units sharded over the ranks, all-gather of the 512-byte partials (RCCL), local combine; and, for
    # N > 1, the no-exchange shard of ranks that hold every share beside it
    if not args.no_extra:
        kt = KEYS["threshold"]["2048"]
        tn, shares = int(kt["n"], 16), [int(s, 16) for s in kt["shares"]]
        ids = [1, 3, 5]                       # every 3-subset of 5 has a negative Lagrange coefficient
        tk = pa.ThresholdPublicKey(ctx, tn, total=5, threshold=3)
        BT = 16384
        rg = np.random.default_rng(4)         # the same ciphertexts on every rank (inputs are public)
        tm_h, tr_h = rand_below(tn, BT, 256, rg), rand_below(tn, BT, 256, rg)
        tr_h[:, -1] |= 1
        tm, tr = torch.from_numpy(tm_h).to(dev), torch.from_numpy(tr_h).to(dev)
        tc = torch.zeros((BT, 512), dtype=torch.uint8, device=dev)
        tk.encrypt_with_r_raw(BT, tm.data_ptr(), 256, tr.data_ptr(), 256, tc.data_ptr(), 512, MEM_DEVICE)
        acc = {"mads": 0.0, "ms": 0.0, "kern": "", "best": 0.0}

        def note():
            pr_ = ctx.last_profile()
            acc["mads"] += pr_["vm_mads"]
            acc["ms"] += pr_["vm_ms"]
            if pr_["vm_mads"] > acc["best"]:
                acc["best"], acc["kern"] = pr_["vm_mads"], pr_["kernel"]

        def partial_fn(s, rows):
            o = torch.empty((rows.shape[0], 512), dtype=torch.uint8, device=dev)
            tk.partial_decrypt_raw(shares[ids[s] - 1], rows.shape[0], rows.data_ptr(), 512, o.data_ptr(), 512, MEM_DEVICE)
            note()
            return o

        def combine_fn(parts):
            o = torch.empty((parts[0].shape[0], 256), dtype=torch.uint8, device=dev)
            tk.combine_raw(ids, parts[0].shape[0], [x.data_ptr() for x in parts], 512, o.data_ptr(), 256, MEM_DEVICE)
            note()
            return o

        def units_fn(server_index, rows):
            # this rank's (server, ciphertext) units in one launch: per-unit exponents (pgpu_partial_decrypt_indexed)
            o = torch.empty((rows.shape[0], 512), dtype=torch.uint8, device=dev)
            rows = rows.contiguous()
            tk.partial_decrypt_indexed_raw([shares[i - 1] for i in ids], server_index, rows.shape[0], rows.data_ptr(), 512,
                                           o.data_ptr(), 512, MEM_DEVICE)
            note()
            return o

        def servers_fn(rows):
            # one rank holding all t shares: the servers' ladders pair up in two-segment launches (pgpu_partial_decrypt_multi)
            outs = [torch.empty((rows.shape[0], 512), dtype=torch.uint8, device=dev) for _ in ids]
            tk.partial_decrypt_multi_raw([shares[i - 1] for i in ids], rows.shape[0], rows.data_ptr(), 512,
                                         [o.data_ptr() for o in outs], 512, MEM_DEVICE)
            note()
            return outs

        def range_fn(rows, ub, ue):
            # this rank's units straight from the ciphertext batch (pgpu_partial_decrypt_units): ciphertexts wanted under several
            # of the rank's shares share one chain of squarings
            o = torch.empty((ue - ub, 512), dtype=torch.uint8, device=dev)
            tk.partial_decrypt_units_raw([shares[i - 1] for i in ids], rows.shape[0], rows.data_ptr(), 512, ub, ue, o.data_ptr(), 512,
                                         MEM_DEVICE)
            note()
            return o

        fns = dict(partial_fn=partial_fn, combine_fn=combine_fn, servers_fn=servers_fn, range_fn=range_fn,
                   # (one launch per rank once a rank's share of a server falls below what fills the chip on its own)
                   units_fn=units_fn if (len(ids) * BT) // world < 32768 else None)
        backend = "gloo (rehearsal)" if args.rehearse_one_gpu else "RCCL"
        # `threshold_2048` is the flow BASELINE config 4 names on every N: the servers' shares on different ranks, unit ranges, ONE
        # all-gather of the partials (RCCL over xGMI), local combine.  With N > 1 the no-exchange shard of a holder of every share is
        # reported beside it (`threshold_2048_replicated`).  Both go through pdist.threshold_step (tests/test_dist_gloo.py calls the same).
        for tname, shard_mode in pdist.threshold_bench_entries(world):
            tmg = {}
            tstep = lambda: pdist.threshold_step(shard_mode, tc, len(ids), rank, world, timings=tmg, **fns)
            tstep()
            barrier()
            acc.update(mads=0.0, ms=0.0, best=0.0)
            tmg.clear()
            t = time.perf_counter()
            for _ in range(args.extra_steps):
                tout, (sb, se) = tstep()
            barrier()
            tel = pdist.max_over_ranks(time.perf_counter() - t, dev)
            tok = pdist.min_over_ranks(1 if (tout is None or torch.equal(tout, tm[sb:se])) else 0, dev)
            if not tok:
                raise SystemExit(f"[bench] threshold decryption ({shard_mode} shard): Combine(PartialDecrypt x 3) != m on some rank")
            exch_ms = pdist.max_over_ranks(tmg.get("exchange_s", 0.0), dev) / args.extra_steps * 1e3
            e = entry(tname, f"t=3 of l=5, servers {ids}, 16384 ciphertexts per step, 2048-bit safe-prime key: 3 x "
                      f"PartialDecrypt + CombinePartialDecryptions; " +
                      (f"ciphertext slices over {world} ranks that hold every share (one chain of squarings per ciphertext, no exchange)"
                       if shard_mode == "ciphertext" else
                       f"(server, ciphertext) units sharded over {world} rank(s), all-gather of the partials" +
                       (f" over {backend}" if world > 1 else " (single rank: no exchange)")) +
                      ", local combine", "threshold decryptions/s", BT, tel / args.extra_steps,
                      acc["ms"] / args.extra_steps, acc["mads"] / args.extra_steps, acc["kern"],
                      "all 16384 plaintexts recovered on every rank", scaling="strong")
            e.update({"scaling": "strong", "n_gpus": world, "shard": shard_mode,
                      "exchange_bytes_per_step": tmg.get("exchange_bytes", 0), "exchange_padded_bytes_per_step": tmg.get("exchange_padded_bytes", 0),
                      "exchange_ms_per_step": exch_ms if shard_mode == "units" and world > 1 else 0.0,
                      "exchange_backend": tmg.get("exchange_backend"), "exchange_world_size": tmg.get("exchange_world", world),
                      "kernel_ms_note": "rank 0's share of the step"})
            if exch_ms and shard_mode == "units" and world > 1:
                e["exchange_GBps"] = tmg.get("exchange_padded_bytes", 0) / (exch_ms * 1e-3) / 1e9
            extras.append(e)
        if world == 1:
            checks["threshold_2048"] = (tn, shares, ids, tc[:64].cpu().numpy(), tm_h[:64])

    # the share-decryption proof (thresholdkey.go:225-326): PartialDecryptionWithZKP (r supplied) and VerifyProof for 16384
    # ciphertexts IN TOTAL under one server's share -- strong scaling: ciphertexts split over the ranks, no exchange
    if not args.no_extra:
        zserver = 2
        zv, zvks = int(kt["v"], 16), [int(x, 16) for x in kt["vks"]]
        zlo, zhi = pdist.shard_slice(BT, rank, world)
        BZ = zhi - zlo
        zr_h = rand_below(tn * tn, BT, 512, np.random.default_rng(44))[zlo:zhi]
        zr, zc = torch.from_numpy(np.ascontiguousarray(zr_h)).to(dev), tc[zlo:zhi].contiguous()
        zdec = torch.zeros((BZ, 512), dtype=torch.uint8, device=dev)
        ze = torch.zeros((BZ, 32), dtype=torch.uint8, device=dev)
        zz = torch.zeros((BZ, 560), dtype=torch.uint8, device=dev)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: tk.share_zkp_prove_raw(
            shares[zserver - 1], zv, BZ, zc.data_ptr(), 512, zr.data_ptr(), 512, zdec.data_ptr(), 512, ze.data_ptr(), zz.data_ptr(), 560,
            MEM_DEVICE)), args.extra_steps)
        extras.append(entry("share_zkp_prove_2048", "16384 PartialDecryptionWithZKP (r supplied), 2048-bit safe-prime key, one server: "
                            "c^(2 delta s), (c^4)^r, V^r mod n^2, SHA-256 of the transcript, Z = r + E delta s (thresholdkey.go:225-257)",
                            "proofs/s", BT, dt, vms, mads, kern, "every proof verifies (below)", scaling="strong"))
        zok = np.zeros(BZ, dtype=np.int32)
        dt, vms, mads, kern = timed(lambda: one_call(lambda: tk.share_zkp_verify_raw(
            zv, zvks[zserver - 1], BZ, zc.data_ptr(), 512, zdec.data_ptr(), 512, ze.data_ptr(), zz.data_ptr(), 560, zok, MEM_DEVICE)),
            args.extra_steps)
        all_ranks_ok(bool(zok.all()), "share ZKP: a proof made by the prover was rejected by VerifyProof")
        extras.append(entry("share_zkp_verify_2048", "16384 VerifyProof, 2048-bit safe-prime key, one server: (c^4)^Z (c_i^2)^-E, "
                            "V^Z v_i^-E mod n^2, SHA-256 of the transcript (thresholdkey.go:278-311)", "proofs/s", BT, dt, vms, mads, kern,
                            "16384 of 16384 accepted", scaling="strong"))
        if world == 1:
            checks["share_zkp_2048"] = (tn, shares[zserver - 1], zv, zvks[zserver - 1], zc[:32].cpu().numpy(), zr_h[:32],
                                        zdec[:32].cpu().numpy(), ze[:32].cpu().numpy(), zz[:32].cpu().numpy())
        del zr, zc, zdec, ze, zz

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    vm_ms_avg = sum(vm_ms) / len(vm_ms)
    value = world * B * args.steps / elapsed
    half = args.bits // 2
    # SURVEY.md §8(d) unit (schoolbook CIOS on 32-bit words, fixed 5-bit window): kept for reference.  The engine needs
    # fewer multiplies than that (squaring symmetry, sliding windows and -- for 2048-bit keys -- residues modulo p^2 as two
    # base-p digits: 3.5 instead of 6 half-width products per squaring), so a rate in SURVEY units can exceed the issue
    # peak.  The roofline therefore prices the kernel in the multiplies its own algorithm needs: 28-bit limb products,
    # counted per VM opcode by the library (squarings H(H-1)/2 + H + 3H^2, products 5H^2 on the pair kernel).
    alg_mul32 = 2 * alg_mul32_per_modexp(args.bits, half)
    alg_mul32_noncrt = alg_mul32_per_modexp(2 * args.bits, args.bits)
    alg_mads = prof["vm_mads"] / B                     # 28-bit multiply-adds the ladder programs need per decryption
    achieved = prof["vm_mads"] / (vm_ms_avg * 1e-3)
    alg_bytes = cb + pb  # read c (n^2 bytes) + write m (n bytes)
    what = {2048: "pair kernel: x^(p-1) mod p^2 and x^(q-1) mod q^2 ladders, both halves in one launch",
            1024: "CRT modexp over p^2 and q^2",
            3072: "one-lane pair kernel (quotient digits in LDS, multiplier digits streamed): ladders modulo p^2 and q^2, both halves in one launch"}[args.bits]
    roofline = {
        "bound": "valu",  # integer multiply issue (v_mad_u64_u32); neither HBM nor MFMA binds
        "kernel": f"{prof['kernel']} ({what})",
        "kernel_name": prof["kernel"],
        "achieved": achieved / 1e12,
        "peak": PEAK_MAD_PER_S / 1e12,
        "unit": "Tmad28/s",
        "frac": achieved / PEAK_MAD_PER_S,
        "kernel_ms_per_launch": vm_ms_avg,
        "alg_mad28_per_decrypt": alg_mads,
        "survey_unit": {"alg_mul32_per_decrypt_crt": alg_mul32, "alg_mul32_per_decrypt_noncrt": alg_mul32_noncrt,
                        "rate_Tmul32_per_s": alg_mul32 * B / (vm_ms_avg * 1e-3) / 1e12,
                        "frac_of_issue_peak": alg_mul32 * B / (vm_ms_avg * 1e-3) / PEAK_MAD_PER_S,
                        "note": "above 1: the engine's algorithm needs fewer multiplies than the schoolbook count of SURVEY 8(d)"},
        # HBM bytes per launch of the dominant kernel from PMC counters (see measure_traffic_pmc): the per-lane window table
        # (written once, one entry read per window product), not re-reads of the inputs
        "traffic": traffic,
        "traffic_source": traffic_note,
        # sustained clock of this kernel's launches in a profiled child pass (attributes slow runs: box clock vs scheduling)
        "clock_ghz": clock["clock_ghz"] if clock else None,
        "clock": clock,
        "hbm": {"achieved": alg_bytes * B / (vm_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": alg_bytes * B / (vm_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, "alg_bytes_per_decrypt": alg_bytes,
                "traffic_GBps": (traffic / (vm_ms_avg * 1e-3) / 1e9) if traffic else None},
    }

    if args.bits == 2048 and B == 65536:
        roofline["plan_deviation"] = plan_check("headline_decrypt_2048", alg_mads)

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        # CPU leg: the libgmp restatement of the reference's call sequence, timed on this box's host cores, and -- with the
        # same library -- the sample checks of the extra configs.  (The reference itself is Go on github.com/ncw/gmp -> libgmp
        # and cannot be built here: no Go toolchain, dependency not vendored; kind = "port".)
        from oracle import gmp_oracle as go
        host = host_cpu_info()
        threads = host["usable_cpus"]
        if host["physical_cores"] and threads > host["physical_cores"]:
            threads = host["physical_cores"]          # mpz_powm saturates a core's multiplier: SMT siblings add nothing
        c_host = c_dev.cpu().numpy()
        go.decrypt_batch_raw(n, lam, c_host[:8], pb, threads=1)          # warm-up: library load, page faults, caches
        t = time.perf_counter()
        out1, _ = go.decrypt_batch_raw(n, lam, c_host[8:40], pb, threads=1)
        per_op = (time.perf_counter() - t) / 32
        go.decrypt_batch_raw(n, lam, c_host[:4 * threads], pb, threads=threads)   # warm the thread pool
        sample = int(max(64, min(B, args.cpu_seconds * threads / per_op)))
        t = time.perf_counter()
        outc, used = go.decrypt_batch_raw(n, lam, c_host[:sample], pb, threads=threads)
        dt = time.perf_counter() - t
        assert (outc == m_host[:sample]).all(), "CPU baseline disagrees with the plaintexts"
        assert (out_dev[:sample].cpu().numpy() == outc).all(), "GPU result differs from the libgmp oracle"
        scrt = min(B, 4 * sample)
        t = time.perf_counter()
        outk, _ = go.decrypt_crt_batch_raw(p, q, c_host[:scrt], pb, threads=threads)
        dtk = time.perf_counter() - t
        assert (outk == m_host[:scrt]).all(), "CRT CPU baseline disagrees with the plaintexts"
        cpu_baseline = {
            "value": sample / dt, "unit": "decryptions/s", "cores": int(used),
            "kind": "port",
            "kind_note": "libgmp restatement of the reference's call sequence; the reference (Go + github.com/ncw/gmp) cannot be "
                         "built in this image: no Go toolchain, dependency not vendored",
            "single_thread_value": 1.0 / per_op,
            "crt_value": scrt / dtk, "crt_note": f"the same harness with textbook CRT over p^2, q^2 (not the reference's "
                                                 f"algorithm), {scrt} ciphertexts, {used} threads",
            "host": host,
            "sample": f"{sample} of the same {args.bits}-bit ciphertexts; libgmp {go.load().oracle_gmp_version().decode()} "
                      f"mpz_powm call sequence of paillier.go:292-303 (no CRT, lambda^-1 per call), {used} OpenMP threads "
                      f"(after a warm-up; single-thread figure from 32 calls after 8 warm-up calls)",
        }
        # sample checks of the extra configs against the same libgmp oracle (+ the Python-int oracle for Combine)
        by = {e["config"]: e for e in extras}
        if "encrypt_2048" in checks:
            nn, mh, rh, ch = checks["encrypt_2048"]
            t = time.perf_counter()
            want, u = go.encrypt_batch_raw(nn, nn + 1, mh, rh, 512, threads=threads)
            by["encrypt_2048"].update({"parity": by["encrypt_2048"]["parity"] + "; 64 ciphertexts == libgmp oracle",
                                       "cpu_per_s": 64 / (time.perf_counter() - t), "cpu_threads": int(u)})
            assert (want == ch).all(), "[bench] Encrypt-2048 differs from the libgmp oracle"
        if "decrypt_3072" in checks:
            nn, ll, ch, mh = checks["decrypt_3072"]
            t = time.perf_counter()
            want, u = go.decrypt_batch_raw(nn, ll, ch, 384, threads=threads)
            by["decrypt_3072"].update({"parity": by["decrypt_3072"]["parity"] + "; 32 plaintexts == libgmp oracle",
                                       "cpu_per_s": 32 / (time.perf_counter() - t), "cpu_threads": int(u)})
            assert (want == mh).all(), "[bench] Decrypt-3072 differs from the libgmp oracle"
        if "ddleq_2048" in checks:
            from paillier_amd.api import be_to_ints
            nn, ll, arrs = checks["ddleq_2048"]
            c1, c2, a_, b_, x_, y_, al_, e_, f_ = [be_to_ints(a) for a in arrs]
            t = time.perf_counter()
            wal, wes, wfs, bits = go.ddleq_prove_batch(nn, ll, c1, c2, a_, b_, x_, y_, threads=threads)
            tp = time.perf_counter() - t
            assert (wal, wes, wfs) == (al_, e_, f_), "[bench] DDLEQ prover differs from the libgmp oracle"
            t = time.perf_counter()
            assert all(go.ddleq_verify_batch(nn, c1, c2, x_, y_, al_, e_, f_, threads=threads))
            tv = time.perf_counter() - t
            by["ddleq_prove_2048"].update({"parity": f"{len(c1)} proofs (challenge bits {bits}) == libgmp oracle; all 16384 verify",
                                           "cpu_per_s": len(c1) / tp, "cpu_threads": min(threads, len(c1))})
            by["ddleq_verify_2048"].update({"parity": by["ddleq_verify_2048"]["parity"] + f"; {len(c1)} verdicts == libgmp oracle",
                                            "cpu_per_s": len(c1) / tv, "cpu_threads": min(threads, len(c1))})
        if "ddleq_secpar40" in checks:
            from paillier_amd.api import be_to_ints
            nn, ll, sp, arrs = checks["ddleq_secpar40"]
            c1, c2, a_, b_, x_, y_, al_, e_, f_ = [be_to_ints(a) for a in arrs]
            t = time.perf_counter()
            wal, wes, wfs, bits = go.ddleq_prove_batch(nn, ll, c1 * sp, c2 * sp, a_ * sp, b_ * sp, x_, y_, threads=threads)
            tp = time.perf_counter() - t
            assert (wal, wes, wfs) == (al_, e_, f_), "[bench] DDLEQ secpar-40 prover differs from the libgmp oracle"
            by["ddleq_prove_2048_secpar40"]["parity"] += f"; the {sp} instances of statement 0 == libgmp oracle (proveDDLEQInstance each)"
            # the reference's ProveDDLEQ (ddleq.go:27-40) runs proveDDLEQInstance secpar times, nothing hoisted: its cost per
            # instance at secpar 40 is the cost of an instance
            by["ddleq_prove_2048_secpar40"].update({"cpu_per_s": sp / tp, "cpu_threads": min(threads, sp),
                                                    "cpu_note": "libgmp proveDDLEQInstance x 40 for one statement (ddleq_test.go:74-88)"})
        if "encrypt_l2_2048" in checks:
            nn, mh, rh, ch = checks["encrypt_l2_2048"]
            t = time.perf_counter()
            want, u = go.encrypt_l2_batch_raw(nn, nn + 1, mh, rh, 768, threads=threads)
            by["encrypt_l2_2048"].update({"parity": by["encrypt_l2_2048"]["parity"] + f"; {len(mh)} ciphertexts == libgmp oracle",
                                          "cpu_per_s": len(mh) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(mh))})
            assert (want == ch).all(), "[bench] level-two Encrypt differs from the libgmp oracle"
        if "decrypt_l2_2048" in checks:
            nn, ll, ch, mh = checks["decrypt_l2_2048"]
            t = time.perf_counter()
            want, u = go.decrypt_l2_batch_raw(nn, ll, ch, 512, threads=threads)
            by["decrypt_l2_2048"].update({"parity": by["decrypt_l2_2048"]["parity"] + f"; {len(ch)} plaintexts == libgmp oracle",
                                          "cpu_per_s": len(ch) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(ch))})
            assert (want == mh).all(), "[bench] level-two Decrypt differs from the libgmp oracle"
        if "nested_add_2048" in checks:
            nn, ch, kh, oh = checks["nested_add_2048"]
            t = time.perf_counter()
            want, u = go.const_mult_batch_raw(nn ** 3, ch, kh, 768, threads=threads)
            by["nested_add_2048"].update({"parity": by["nested_add_2048"]["parity"] + f"; {len(ch)} ciphertexts == libgmp oracle",
                                          "cpu_per_s": len(ch) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(ch))})
            assert (want == oh).all(), "[bench] NestedAdd differs from the libgmp oracle"
        if "nested_randomize_2048" in checks:
            nn, (c1h, ah, bh, c2h) = checks["nested_randomize_2048"]
            t = time.perf_counter()
            want, u = go.nested_randomize_batch_raw(nn, c1h, ah, bh, threads=threads)
            by["nested_randomize_2048"].update({"parity": f"{len(c1h)} ciphertexts == libgmp oracle; " + by["nested_randomize_2048"]["parity"],
                                                "cpu_per_s": len(c1h) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(c1h))})
            assert (want == c2h).all(), "[bench] NestedRandomize differs from the libgmp oracle"
        if "homomorphic_2048" in checks:
            nn, k50_, ah, bh, cmh, add_rows_, sub_rows_ = checks["homomorphic_2048"]
            reps = 64                                       # (an Add is ~2 us of libgmp: repeat the 64 pairs so that the clock sees it)
            abig, bbig = np.tile(ah, (reps, 1)), np.tile(bh, (reps, 1))
            go.add_sub_batch_raw(nn * nn, False, abig, bbig, 512, threads=threads)
            t = time.perf_counter()
            wadd, u, _ = go.add_sub_batch_raw(nn * nn, False, abig, bbig, 512, threads=threads)
            ta = time.perf_counter() - t
            t = time.perf_counter()
            wsub, u2, oks = go.add_sub_batch_raw(nn * nn, True, wadd, bbig, 512, threads=threads)
            ts = time.perf_counter() - t
            assert (wsub == abig).all() and oks.all(), "[bench] libgmp Sub(Add(a, b), b) != a"
            assert (wadd[:64] == add_rows_).all() and (wsub[:64] == sub_rows_).all(), "[bench] Add / Sub differ from the libgmp oracle"
            for nm_ in ("add_2048", "sub_2048"):
                by[nm_]["parity"] += "; 64 rows == libgmp oracle"
            by["add_2048"].update({"cpu_per_s": len(abig) / ta, "cpu_threads": int(u),
                                   "cpu_note": "libgmp Mul + Mod per operand (operations.go:18-21), 4096 pairs"})
            by["sub_2048"].update({"cpu_per_s": len(abig) / ts, "cpu_threads": int(u2),
                                   "cpu_note": "libgmp ModInverse + Mul + Mod (operations.go:43-47), 4096 pairs"})
            t = time.perf_counter()
            want, u = go.const_mult_batch_raw(nn * nn, ah, k50_, 512, threads=threads)
            by["const_mult_2048"].update({"parity": by["const_mult_2048"]["parity"] + "; 64 ciphertexts == libgmp oracle",
                                          "cpu_per_s": len(ah) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(ah))})
            assert (want == cmh).all(), "[bench] ConstMult differs from the libgmp oracle"
        if "const_mult_full_2048" in checks:
            nn, ch, kh, oh = checks["const_mult_full_2048"]
            t = time.perf_counter()
            want, u = go.const_mult_batch_raw(nn * nn, ch, kh, 512, threads=threads)
            by["const_mult_2048_full_k"].update({"parity": by["const_mult_2048_full_k"]["parity"] + f"; {len(ch)} ciphertexts == libgmp oracle",
                                                 "cpu_per_s": len(ch) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(ch))})
            assert (want == oh).all(), "[bench] ConstMult (full-width k) differs from the libgmp oracle"
        if "alt_encrypt_2048" in checks:
            nn, hh, kk_, mh, rh, ch = checks["alt_encrypt_2048"]
            t = time.perf_counter()
            want, u, _ = go.alt_encrypt_batch_raw(nn, nn + 1, hh, kk_, mh, rh, 512, threads=threads)
            by["alt_encrypt_2048"].update({"parity": by["alt_encrypt_2048"]["parity"] + f"; {len(mh)} ciphertexts == libgmp oracle",
                                           "cpu_per_s": len(mh) / (time.perf_counter() - t), "cpu_threads": min(int(u), len(mh))})
            assert (want == ch).all(), "[bench] AltEncrypt differs from the libgmp oracle"
        if "share_zkp_2048" in checks:
            nn, sh_, v_, vi_, ch, rh, dh, eh, zh = checks["share_zkp_2048"]
            t = time.perf_counter()
            wd, we_, wz, u = go.share_zkp_prove_batch_raw(nn, 5, sh_, v_, ch, rh, 560, threads=threads)
            tp = time.perf_counter() - t
            assert (wd == dh).all() and (we_ == eh).all() and (wz == zh).all(), "[bench] share ZKP prover differs from the libgmp oracle"
            t = time.perf_counter()
            wok, u2 = go.share_zkp_verify_batch_raw(nn, v_, vi_, ch, dh, eh, zh, threads=threads)
            tv = time.perf_counter() - t
            assert wok.all(), "[bench] libgmp VerifyProof rejects a GPU proof"
            by["share_zkp_prove_2048"].update({"parity": f"{len(ch)} proofs (Decryption, E, Z) == libgmp oracle; all 16384 verify",
                                               "cpu_per_s": len(ch) / tp, "cpu_threads": min(int(u), len(ch))})
            by["share_zkp_verify_2048"].update({"parity": by["share_zkp_verify_2048"]["parity"] + f"; {len(ch)} verdicts == libgmp oracle",
                                                "cpu_per_s": len(ch) / tv, "cpu_threads": min(int(u2), len(ch))})
        if "threshold_2048" in checks:
            from oracle import paillier_oracle as po
            from paillier_amd.api import be_to_ints
            tn_, sh, ids_, ch, mh = checks["threshold_2048"]
            tsks = [po.ThresholdSecretKey(N=tn_, G=tn_ + 1, TotalNumberOfDecryptionServers=5, Threshold=3, ID=i, Share=sh[i - 1])
                    for i in ids_]
            got = [po.combine_partial_decryptions(tsks[0], [po.partial_decrypt(ts, c) for ts in tsks]) for c in be_to_ints(ch[:4])]
            assert got == be_to_ints(mh[:4]), "[bench] threshold decryption differs from the oracle"
            # the shape of the reference's own benchmark (thresholdkey_test.go:396-427): per ciphertext 3 x PartialDecrypt +
            # CombinePartialDecryptions, libgmp call for call
            t = time.perf_counter()
            want, u = go.threshold_decrypt_batch_raw(tn_, 5, ids_, [sh[i - 1] for i in ids_], ch, 256, threads=threads)
            tt = time.perf_counter() - t
            assert (want == mh).all(), "[bench] threshold decryption differs from the libgmp oracle"
            by["threshold_2048"].update({"parity": by["threshold_2048"]["parity"] + f"; {len(ch)} ciphertexts: libgmp PartialDecrypt x 3 + "
                                         "Combine == m (4 of them also through the Python-int oracle)",
                                         "cpu_per_s": len(ch) / tt, "cpu_threads": min(int(u), len(ch))})

    line = {
        "metric": f"paillier_{args.bits}bit_decryptions_per_s",
        "value": value,
        "unit": "decryptions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (28-bit limbs, 64-bit accumulators)",
        "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if args.rehearse_one_gpu else ""),
        "bit_exact": ok,
        "config": {"workload": f"Batch {B} Decrypt per GPU, {args.bits}-bit n, level 1, CRT over p^2,q^2; inputs resident "
                               f"in HBM as big-endian element-major bytes",
                   "batch_per_gpu": B, "key_bits": args.bits, "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "world_size": world},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        # True: some config executed a different number of multiplies per unit than profiles/plan_table.json records (+- 1 %) --
        # a planning predicate moved it onto another ladder; its numbers are then NOT comparable with earlier rounds'
        "plan_changed": bool(plan_dev),
        "plan_deviations": plan_dev,
        "encrypt_setup": {"vm_ms": enc_prof["vm_ms"], "encryptions_per_s": B / (enc_prof["vm_ms"] * 1e-3)},
        "extra_configs": extras,
    }
    if plan_dev:
        print("[bench] PLAN CHANGED (executed multiply-adds per unit differ from profiles/plan_table.json): " +
              ", ".join(f"{k} {v['deviation']:+.1%}" for k, v in plan_dev.items()), file=sys.stderr)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
