#!/usr/bin/env python3
"""tools/pmc_k5.sh's passes c and d (wait cycles by cause, vector-memory writes, the texture cache's pending-request stalls and
L2 request latencies) of the register-cached K5 -> profiles/<tag>_k5_wait.md: which queue the waves sit in.

    python tools/k5_wait_md.py gpurun_out/pmc_k5_<tag> <tag>
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_dispatch(out_dir, sub):
    fs = glob.glob(f"{out_dir}/{sub}/**/*counter_collection.csv", recursive=True)
    agg, n = collections.defaultdict(float), set()
    if not fs:
        return {}, 0
    for r in csv.DictReader(open(fs[0])):
        if "shot_cached" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        n.add(r["Dispatch_Id"])
    return {k: v / max(len(n), 1) for k, v in agg.items()}, len(n)


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    from shot_fpfh_amd import _ffi

    build = _ffi.load().sf_version().decode()
    a, _ = per_dispatch(out_dir, "a")
    c, nc = per_dispatch(out_dir, "c")
    d, nd = per_dispatch(out_dir, "d")
    for sub in ("e", "f"):  # (the texture cache's counters come two per pass)
        more, _ = per_dispatch(out_dir, sub)
        for k, v in more.items():
            d.setdefault(k, v)
    lines = [f"# K5 (`k_shot_cached`): what its waves wait for ({tag}, build {build})", "",
             "`tools/pmc_k5.sh` (rocprofv3 --pmc, separate passes, SHOT chain alone at C3: one wave per keypoint), per wave "
             "(= per keypoint); SQ cycle counters are in units of 4 cycles.", ""]
    w = c.get("SQ_WAVES") or a.get("SQ_WAVES") or 1.0
    if c:
        life = c.get("SQ_WAVE_CYCLES", 0) / w
        rows = [("wave lifetime (SQ_WAVE_CYCLES x 4)", 4 * life),
                ("waiting for anything (SQ_WAIT_ANY x 4)", 4 * c.get("SQ_WAIT_ANY", 0) / w),
                ("waiting for an instruction to issue (SQ_WAIT_INST_ANY x 4)", 4 * c.get("SQ_WAIT_INST_ANY", 0) / w),
                ("... of which behind an LDS instruction (SQ_WAIT_INST_LDS x 4)", 4 * c.get("SQ_WAIT_INST_LDS", 0) / w),
                ("issuing any instruction (SQ_ACTIVE_INST_ANY x 4)", 4 * c.get("SQ_ACTIVE_INST_ANY", 0) / w)]
        lines += ["| per wave | cycles | share of the lifetime |", "|---|---|---|"]
        lines += [f"| {n} | {v:,.0f} | {v / (4 * life) if life else 0:.2f} |" for n, v in rows]
        lines += ["", f"Vector-memory instructions per wave: {c.get('SQ_INSTS_VMEM_RD', 0) / w:.1f} reads, {c.get('SQ_INSTS_VMEM_WR', 0) / w:.1f} "
                  f"writes (the 2.8 KB row: three 16-byte-per-lane stores; the frame write-back when a sign flipped).", ""]
    if d:
        wd = d.get("SQ_WAVES") or w
        rd, wr = d.get("TCP_TCC_READ_REQ_sum", 0), d.get("TCP_TCC_WRITE_REQ_sum", 0)
        lines += ["| texture cache (TCP), per wave | value |", "|---|---|",
                  f"| read requests to the L2 | {rd / wd:.1f} |", f"| write requests to the L2 | {wr / wd:.1f} |",
                  f"| mean latency of a read request (TCP_TCC_READ_REQ_LATENCY / requests) | {d.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / rd if rd else 0:,.0f} cycles |",
                  f"| mean latency of a write request (TCP_TCC_WRITE_REQ_LATENCY / requests) | {d.get('TCP_TCC_WRITE_REQ_LATENCY_sum', 0) / wr if wr else 0:,.0f} cycles |",
                  f"| cycles stalled on pending requests (TCP_PENDING_STALL_CYCLES) | {d.get('TCP_PENDING_STALL_CYCLES_sum', 0) / wd:,.0f} |", ""]
    lines += [f"Dispatches averaged: {nc} (pass c), {nd} (pass d).  Raw counters: `gpurun_out/pmc_k5_{tag}/` on the box that ran them.", ""]
    path = os.path.join(ROOT, "profiles", f"{tag}_k5_wait.md")
    open(path, "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
