#!/usr/bin/env python3
"""gpurun_out/<round>/* (tools/round_profiles.sh) -> profiles/<round>_* (round tag: argv[1], default r03): kernel stats of the library's kernels, the decoder's launches, one line per
kernel and counter pass, and the traffic json bench.py quotes."""
import ast, csv, glob, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(R, "gpurun_out", RD); dst = os.path.join(R, "profiles")
rows = list(csv.reader(open(os.path.join(src, "bench_kernel_stats.csv"))))
with open(os.path.join(dst, RD + "_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if "dsr::" in r[0]:
            w.writerow(r)
open(os.path.join(dst, RD + "_bench_viterbi_launches.txt"), "w").write(open(os.path.join(src, "bench_viterbi_launches.txt")).read())
line = [l for l in open(os.path.join(src, "bench_traced.json")) if l.startswith("{")][-1]
open(os.path.join(dst, RD + "_bench_traced.json"), "w").write(line)
acc = {}
for fn in sorted(glob.glob(os.path.join(src, "pmc_*.txt"))):
    for l in open(fn):
        m = re.match(r"(.*?) (\{.*\}) launches (\d+)", l.strip())
        if not m or "dsr::" not in m.group(1):
            continue
        k = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("dsr::", "")
        d = ast.literal_eval(m.group(2))
        acc.setdefault(k, {"launches": int(m.group(3))}).update({c: float(v) for c, v in d.items()})
with open(os.path.join(dst, RD + "_pmc_summary.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <one set per pass> -- python3 bench.py --serial --steps 1 --warmup 1 --no-cpu --beam 53.787 (tools/round_profiles.sh)\n")
    f.write("# counters summed over the launches of the run (decoder: 4-utterance probe + 2 full batches; other kernels: probe + 2)\n")
    for k in sorted(acc):
        f.write(k + " " + json.dumps(acc[k], sort_keys=True) + "\n")
    v = acc.get("k_viterbi") or acc.get("k_viterbi<0>")
    if v:
        wf = 2 * 1000 * 1000 * 16.0          # wave-frames of the two full batches (the probe adds 0.2 %)
        f.write("\n# k_viterbi per wave and frame: VALU %.0f  SALU %.0f  LDS %.0f;  wait-any %.0f %%  wait-for-issue %.0f %% of the wave cycles;  LDS bank conflicts %.0f %% of the LDS cycles\n"
                % (v["SQ_INSTS_VALU"] / wf, v["SQ_INSTS_SALU"] / wf, v["SQ_INSTS_LDS"] / wf, 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                   100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    for gk in [k for k in acc if k.startswith("k_gmm_mfma_sp") or k.startswith("k_gmm_mfma_reg")]:
        g = acc.get(gk)
        if g and g.get("SQ_BUSY_CU_CYCLES"):
            f.write("# %s: MFMA busy %.0f %% of the SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / 4 / SQ_BUSY_CU_CYCLES)\n" % (gk.split("<")[0], 100 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / 4 / g["SQ_BUSY_CU_CYCLES"]))
tj = {"config": "bench.py --serial --steps 1 --warmup 1 --no-cpu --beam 53.787 (1000 utt x 10 s x 8 ch per GPU); counters summed over the probe (4 utterances) and two full-batch launches; per-launch = sum / 2",
      "beam": 53.787,
      "collection": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/round_profiles.sh); unit KB", "kernels": {}}
notes = {"k_viterbi": "FETCH_SIZE is RDREQ x 64 B; 128-B requests are tallied at 64 B on gfx950 and this kernel's mixed 4/8/16/32-B gathers are not a calibrated pattern: the read part is a lower bound (true value between 1x and 2x)",
         "k_analysis_q256<4, 2>": "16-B/lane streaming reads: FETCH_SIZE doubled (guide, HBM section)",
         "k_analysis_bf_q256<4>": "the pipe's fused analysis bank + beamformer: 16-B/lane streaming reads of the samples (each tile re-reads the 7 blocks it shares with the tile before), FETCH_SIZE doubled (guide, HBM section); the channel snapshots are not written",
         "k_bf_apply": "8-B/lane streaming reads, whole 128-B requests: FETCH_SIZE doubled (guide, HBM section)"}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch = v["FETCH_SIZE"] * (2.0 if (k.startswith("k_analysis_q256") or k.startswith("k_analysis_bf_q256") or k.startswith("k_bf_apply")) else 1.0)
        name = k.split("<")[0]
        tj["kernels"][name] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "bytes_per_launch": (fetch + v["WRITE_SIZE"]) * 1024.0 / 2.0}
        if k in notes or name in notes:
            tj["kernels"][name]["note"] = notes.get(k) or notes[name]
json.dump(tj, open(os.path.join(dst, RD + "_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, RD + "_pmc_summary.txt")).read()[-900:])
