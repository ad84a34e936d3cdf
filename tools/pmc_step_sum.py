"""HBM traffic of one bench step from the PMC passes: sum over kernels of (FETCH_SIZE + WRITE_SIZE per launch, KB,
tools/pmc_summary.py) x launches per step (tools/kstats.py table); set-up kernels (k_yak_insert, k_encode_ref) left out.
   python tools/pmc_step_sum.py profiles/r04_yeast_one_group_pmc_fetch_write.json profiles/r04_yeast_one_group_kernels_per_step.txt"""
import json, sys
d = json.load(open(sys.argv[1]))
calls = {}
for l in open(sys.argv[2]):
    p = l.split()
    if len(p) > 4 and p[1] == "calls/step":
        calls[p[0].split("<")[0]] = calls.get(p[0].split("<")[0], 0.0) + float(p[2])
rows, tot_f, tot_w = [], 0.0, 0.0
for k, v in d.items():
    name = k.replace("np2::", "").split("<")[0]
    if name not in calls or name in ("k_yak_insert", "k_encode_ref", "k_pack_ref", "k_columnarise"):
        continue
    f, w = v.get("FETCH_SIZE_KB_avg_per_launch", 0.0), v.get("WRITE_SIZE_KB_avg_per_launch", 0.0)
    rows.append(((f + w) * calls[name] / 1024, name, f * calls[name] / 1024, w * calls[name] / 1024, calls[name]))
    tot_f += f * calls[name]
    tot_w += w * calls[name]
rows.sort(reverse=True)
print(f"{'kernel':28s} {'MB/step':>9s} {'fetch':>9s} {'write':>9s} {'calls':>6s}")
for t, n, f, w, c in rows[:24]:
    print(f"{n:28s} {t:9.1f} {f:9.1f} {w:9.1f} {c:6.1f}")
print(f"sum: {(tot_f + tot_w) * 1024 / 1e9:.2f} GB raw per step (fetch {tot_f * 1024 / 1e9:.2f} + write {tot_w * 1024 / 1e9:.2f}); "
      f"with the guide's 2 x FETCH: {(2 * tot_f + tot_w) * 1024 / 1e9:.2f} GB")
