"""End to end from FILES at BASELINE configs[3] scale: a 248 Mb diploid contig's coordinate-sorted BAM (+ .bai), the
assembly FASTA and two .yak dumps padded to >= 10^9 words each on local disk -> `nextPolish2` (nextpolish2_amd.cli,
in process, one rank) -> FASTA file; then the same under torchrun with 2 ranks on this one GPU (--shard_min_len: the contig
cut into 2 reference intervals, every rank reading only its interval's records; gloo).  Reports the stages the command
line prints (NP2_CLI_PROFILE / NP2_IO_PROFILE: yak load, front end — inflate / record walk / columnarise —, polish,
write-out), CPU seconds, and compares the FASTA with the resident path.
   python tools/e2e_chr1_probe.py [L] [table_words]       (defaults 248e6, 1e9)"""
import os, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nextpolish2_amd import Opts, Polisher, cli
from nextpolish2_amd import io as np2io
from nextpolish2_amd._cpus import usable_cpus
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth, concat_pileups
from test_gpu_t2t_share import pad_table

L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_000_000
WORDS = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
NP = 16
log = lambda s: print(s, flush=True)
t = time.time()
with ThreadPoolExecutor(NP) as ex:
    parts = list(ex.map(lambda i: Synth(L // NP, depth=30, seed=500 + i, diploid=True), range(NP)))
pu = concat_pileups([p.pileup for p in parts], "chr1")
log(f"generated in {time.time() - t:.1f} s: L {pu.L}, {pu.n_reads} reads, {pu.n_columns() / 1e9:.2f} G columns")
td = tempfile.mkdtemp(prefix="np2_e2e_", dir=os.environ.get("NP2_E2E_TMP", "/tmp"))
t = time.time()
rng = np.random.default_rng(4)
yaks, ypaths = [], []
for k in (21, 31):
    y = pad_table(Synth.yak_assembly(parts, k, threads=usable_cpus()), WORDS, rng)
    ypaths.append(os.path.join(td, f"k{k}.yak"))
    np2io.write_yak(ypaths[-1], y)
    yaks.append(y)
log(f"yak dumps: {[len(y.words) for y in yaks]} words, {sum(os.path.getsize(p) for p in ypaths) / 2**30:.1f} GiB written in {time.time() - t:.1f} s")
t = time.time()
blobs, offs, poss, rls = [], [np.zeros(1, dtype=np.uint64)], [], []
pos0, name0, base = 0, 0, 0
for p in parts:  # (reads never span a joint: the pieces' sorted record lists, end to end, are the contig's sorted list)
    b, o, ps, rl = p.bam_records(0, pos0, name0)
    blobs.append(b)
    offs.append(o[1:] + np.uint64(base))
    poss.append(ps)
    rls.append(rl)
    base += len(b)
    pos0 += p.pileup.L
    name0 += len(ps)
bam = os.path.join(td, "chr1.bam")
write_bam_raw(bam, [("chr1", pu.L)], [(b"".join(blobs), np.concatenate(offs), np.concatenate(poss), np.concatenate(rls))],
              threads=max(16, usable_cpus()))
del blobs
fa = os.path.join(td, "chr1.fa")
with open(fa, "wb") as f:
    f.write(b">chr1\n" + pu.ref.tobytes() + b"\n")
log(f"BAM {os.path.getsize(bam) / 2**30:.2f} GiB (+ .bai), FASTA written in {time.time() - t:.1f} s")
del parts
env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", NP2_CLI_PROFILE="1", NP2_IO_PROFILE="1")
first_outs = []
if os.environ.get("NP2_E2E_FRESH_FIRST"):
    # Before this process has touched the GPU: the command as the first user of the device, then again at once (the previous
    # process's ~58 GB of HBM were freed a moment ago), then after a pause.  NP2_E2E_THREADS: its -t.
    # NP2_E2E_FRESH_FIRST = "pause:threads,..." (default: at once, at once, after 12 s, at once; -t 2)
    spec = os.environ["NP2_E2E_FRESH_FIRST"]
    # an entry may go on with ":mode:cpus" — NP2_INFLATE for that process (gpu / libdeflate / zlib; empty: the default rule)
    # and a taskset CPU list (a rank's share of a node: "0-1")
    plan = [(0, "2", "", ""), (0, "2", "", ""), (12, "2", "", ""), (0, "2", "", "")] if spec == "1" else \
        [(int(f[0]), f[1], f[2] if len(f) > 2 else "", f[3] if len(f) > 3 else "") for f in (x.split(":") for x in spec.split(","))]
    for rep, (pause, nt, mode, cpus) in enumerate(plan):
        time.sleep(pause)
        o = os.path.join(td, f"first{rep}.fa")
        t = time.time()
        e2 = dict(env, **({"NP2_INFLATE": mode} if mode else {}))
        pre = ["taskset", "-c", cpus] if cpus else []
        r = subprocess.run(pre + [sys.executable, "-m", "nextpolish2_amd.cli", "-t", nt, "-o", o, bam, fa] + ypaths, env=e2, cwd=ROOT,
                           capture_output=True, timeout=900)
        dt = time.time() - t
        sys.stderr.write(r.stderr.decode()[-4000:])
        sys.stderr.flush()
        log(f"fresh process {rep} (-t {nt}, NP2_INFLATE={mode or 'default'}, CPUs {cpus or 'all'}, started {pause} s after the previous one ended): "
            f"{dt:.2f} s = {pu.L / dt / 1e6:.0f} Mbp/s, rc {r.returncode}")
        first_outs.append(o)
# the resident path's answer
t = time.time()
pol = Polisher(yaks)
c = pol.upload(pu)
b, span = pol.polish_resident(c, Opts(), want_pos=False)
want = b">chr1 start:%d end:%d\n%s\n" % (span[0], span[1], b.tobytes())
c.free()
pol.close()
del pol, yaks
log(f"resident path (tables, upload, polish): {time.time() - t:.1f} s")
for o in first_outs:
    log(f"{os.path.basename(o)} == resident path: {open(o, 'rb').read() == want}")
if os.environ.get("NP2_E2E_ONLY_FRESH"):
    sys.exit(0)
os.environ["NP2_CLI_PROFILE"] = "1"
os.environ["NP2_IO_PROFILE"] = "1"
for rep in range(0 if os.environ.get("NP2_E2E_ONLY_2RANK") else 2):
    out = os.path.join(td, f"one{rep}.fa")
    t = time.time()
    rc = cli.main([bam, fa] + ypaths + ["-o", out, "-t", "2"])
    dt = time.time() - t
    same = open(out, "rb").read() == want
    log(f"run {rep}: nextPolish2 (one rank) {dt:.2f} s = {pu.L / dt / 1e6:.0f} Mbp/s, rc {rc}, FASTA == resident path: {same}")
env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
# the same command as a process of its own (what a user runs: interpreter + HIP start-up included, nothing warm)
out1 = os.path.join(td, "fresh.fa")
t = time.time()
r = subprocess.run(["true"] if os.environ.get("NP2_E2E_ONLY_2RANK") else [sys.executable, "-m", "nextpolish2_amd.cli", "-t", "2", "-o", out1, bam, fa] + ypaths, env=env, cwd=ROOT,
                   capture_output=True, timeout=900)
dt = time.time() - t
sys.stderr.write(r.stderr.decode()[-4000:])
if not os.environ.get("NP2_E2E_ONLY_2RANK"):
    log(f"fresh process: nextPolish2 {dt:.2f} s = {pu.L / dt / 1e6:.0f} Mbp/s, rc {r.returncode}, FASTA == resident path: "
        f"{r.returncode == 0 and open(out1, 'rb').read() == want}")
trace = os.environ.get("NP2_E2E_TRACE")
if trace:  # the same process under rocprofv3 (kernel + HIP API timeline: tools/trace_stalls.py reads it)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--hip-trace", "--output-format", "csv", "-d", trace, "--", sys.executable, "-m",
                        "nextpolish2_amd.cli", "-t", "2", "-o", os.path.join(td, "traced.fa"), bam, fa] + ypaths, env=env, cwd=ROOT,
                       capture_output=True, timeout=900)
    sys.stderr.write(r.stderr.decode()[-3000:])
    log(f"traced run rc {r.returncode}")
if os.environ.get("NP2_E2E_SKIP_2RANK"):
    sys.exit(0)
out2 = os.path.join(td, "two.fa")
t = time.time()
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29731", "-m", "nextpolish2_amd.cli", "--dist_backend", "gloo", "--device", "0",
                    "--shard_min_len", "1000000", "-o", out2, bam, fa] + ypaths, env=env, cwd=ROOT, capture_output=True, timeout=1500)
dt = time.time() - t
sys.stderr.write(r.stderr.decode()[-3000:])
log(f"2 ranks on this GPU (2 reference intervals, each rank reads its interval's records; gloo; process start included): "
    f"{dt:.1f} s, rc {r.returncode}, FASTA == resident path: {r.returncode == 0 and open(out2, 'rb').read() == want}")
