"""profiles/<tag>_step_sq.md from tools/pmc_step.sh: what the SIMDs do during every kernel of the bench step (config 3).
python tools/step_sq_md.py gpurun_out/pmc_step_<tag> <tag>"""
import collections, csv, glob, os, re, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = {}
for sub in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            name = m.group(0)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVES":
                agg[name]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    data[sub] = agg
sys.path.insert(0, ROOT)
try:
    from shot_fpfh_amd import _ffi
    build = _ffi.load().sf_version().decode()
except Exception as e:  # noqa: BLE001
    build = f"unknown ({e})"
mean = lambda c, k: sum(c[k]) / len(c[k]) if c.get(k) else float("nan")  # noqa: E731
out = [f"# What the SIMDs do during each kernel of the bench step ({tag}, {build})", "",
       "`tools/pmc_step.sh`: two `rocprofv3 --pmc` passes of `bench.py` (1M uniform points, all keypoints, radius 0.03; dispatches "
       "serialised by the profiler, a few per cent slower than untraced).  Per launch, averaged over the launches of the run; SQ cycle "
       "counters are in units of 4 cycles.  `vector issue` = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x duration x clock): the share of "
       "the kernel's time its SIMDs spend issuing vector instructions; `waves / SIMD` = SQ_WAVE_CYCLES x 4 over the same denominator; "
       "`matrix cores` = SQ_VALU_MFMA_BUSY_CYCLES over the same denominator; `LDS` = SQ_ACTIVE_INST_LDS x 4 over it.", "",
       "| kernel | µs | waves | vector instr. / wave | scalar instr. / wave | vector issue | waves / SIMD in flight | waiting for an instruction (share of wave lifetime) | LDS | LDS bank-conflict cycles / LDS cycles | matrix cores | vector-memory reads + writes / wave |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|"]
rows = []
for name, a in data["a"].items():
    us = mean(a, "_us")
    if not us == us or us < 5:
        continue
    b = data["b"].get(name, {})
    clock = mean(a, "GRBM_GUI_ACTIVE") / 8 / us  # MHz
    den = 1024 * us * clock
    w = mean(a, "SQ_WAVES")
    usb = mean(b, "_us") if b else float("nan")
    denb = 1024 * usb * (mean(b, "GRBM_GUI_ACTIVE") / 8 / usb) if b and usb == usb else float("nan")
    wb = mean(b, "SQ_WAVES") if b else float("nan")
    rows.append((us, f"| `{name}` | {us:.0f} | {w:,.0f} | {mean(a, 'SQ_INSTS_VALU') / w:.0f} | {mean(a, 'SQ_INSTS_SALU') / w:.0f} | "
                     f"{mean(a, 'SQ_ACTIVE_INST_VALU') * 4 / den:.2f} | {mean(a, 'SQ_WAVE_CYCLES') * 4 / den:.1f} | "
                     f"{mean(a, 'SQ_WAIT_INST_ANY') / max(mean(a, 'SQ_WAVE_CYCLES'), 1):.2f} | "
                     f"{mean(b, 'SQ_ACTIVE_INST_LDS') * 4 / denb:.2f} | {mean(b, 'SQ_LDS_BANK_CONFLICT') / max(mean(b, 'SQ_ACTIVE_INST_LDS') * 4, 1):.2f} | "
                     f"{mean(b, 'SQ_VALU_MFMA_BUSY_CYCLES') / denb:.2f} | {(mean(b, 'SQ_INSTS_VMEM_RD') + mean(b, 'SQ_INSTS_VMEM_WR')) / wb:.1f} |"))
out += [r for _, r in sorted(rows, reverse=True)]
path = os.path.join(ROOT, "profiles", f"{tag}_step_sq.md")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
