"""BASELINE.json configs[4] — a CHM13-sized (3.05 Gb) T2T assembly, 30x HiFi, k21 + k31 tables, -r on, 8 GPUs — as ONE
GPU sees it: rank 0's share of the chromosomes (whole contigs, longest first: dist.assign_contigs(lengths, 8)[0],
~380 Mb with chr1 in it), polished with use_all_reads against k-mer tables padded to human scale (>= 10^9 words each).
The shorter chromosome of the share is DIPLOID (15x + 15x reads of two haplotypes), so that -r has something to
switch: heterozygous regions exist, reads that disagree with the contig's own candidate are kept in the read graph
(main.rs:976) and stay in the pileup unless the Louvain vote removes them (main.rs:1004-1010).
Checked through size-independent properties — every haploid contig equals the simulated truth, a second run is
byte-identical, positions never decrease, the diploid chromosome's phasing vote removes reads — and, for one
generated piece of the diploid chromosome (>= 1 Mb, polished as a contig of its own against the SAME human-sized
tables), stage by stage against the oracle under -r (hete.*, invalid_ids, every other trace, the final sequence).
The suite runs at full size (NP2_T2T_SCALE=1: 348 Mb, tables of 10^9 words each); NP2_T2T_SCALE=0.1 is a quick variant;
tools/t2t_share_probe.py logs the device memory high-water mark."""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd._types import Yak
from nextpolish2_amd.dist import assign_contigs
from nextpolish2_amd.synth import Synth, concat_pileups

pytestmark = pytest.mark.gpu

# T2T-CHM13 v2.0 chromosome lengths (bp): chr1..chr22, chrX
CHM13 = [248387328, 242696752, 201105948, 193574945, 182045439, 172126628, 160567428, 146259331, 150617247, 134758134,
         135127769, 133324548, 113566686, 101161492, 99753195, 96330374, 84276897, 80542538, 61707364, 66210255, 45090682,
         51324926, 154259566]


def pad_table(y, n_words, rng):
    """`y` with fabricated words added until it holds n_words, spread evenly over the 1024 file buckets.  Their keys lie
    above every real key of that k (a k-mer hash has 2k bits: key = hash >> 10 < 2^(2k - 10)), so no lookup ever finds
    them: they only fill the HBM table to the size a human read set gives it."""
    k = y.k
    extra = max(0, n_words - len(y.words))
    per = extra // 1024
    if per == 0:
        return y
    lo = np.uint64(1) << np.uint64(2 * k - 10)
    # distinct inside a bucket AND scattered like hash values (i * odd mod 2^31 is a bijection, also on its low bits): a
    # run of consecutive keys would be one solid cluster in a table probed linearly, every real key behind it walking its
    # whole length
    pad_keys = lo + ((np.arange(per, dtype=np.uint64) * np.uint64(0x9E3779B1)) & np.uint64((1 << 31) - 1))
    out = np.empty(len(y.words) + per * 1024, dtype=np.uint64)
    off = np.zeros(1025, dtype=np.uint64)
    w = 0
    for b in range(1024):
        a, e = int(y.bucket_off[b]), int(y.bucket_off[b + 1])
        out[w:w + e - a] = y.words[a:e]
        w += e - a
        out[w:w + per] = (pad_keys << np.uint64(10)) | rng.integers(5, 1000, size=per, dtype=np.uint64)
        w += per
        off[b + 1] = w
    return Yak(k, out, off)


STAGES = ["graph.off", "graph.bases", "graph.delta", "graph.count", "cns_raw.pos", "cns_raw.base", "lq.start", "lq.end",
          "cand.cand_off", "cand.order", "cand.seq_off", "cand.seq", "cand.kmer", "cand.kscore", "hete.lable",
          "hete.kscore", "invalid_ids", "seed.lable", "seed.sudo", "seed.cand_off", "seed.order", "cns_succ.pos",
          "cns_succ.base", "rech0.kscore", "rech0.lable", "rech0.sudo", "cns_rech0.pos", "cns_rech0.base",
          "rech1.kscore", "rech1.lable", "rech1.sudo", "cns_rech1.pos", "cns_rech1.base"]


def run_share(scale, table_words, log=print):
    import torch
    from nextpolish2_amd._cpus import usable_cpus
    from oracle.np2_oracle import Oracle
    lengths = [max(200_000, int(l * scale)) for l in CHM13]
    mine = assign_contigs(lengths, 8)[0]
    share = [lengths[i] for i in mine]
    dip_ci = mine[-1] if len(mine) > 1 else -1  # the shorter chromosome of the share is diploid
    log(f"rank 0 of 8 polishes chromosomes {[i + 1 for i in mine]}: {sum(share) / 1e6:.1f} Mb of {sum(lengths) / 1e6:.1f} Mb; "
        f"chr{dip_ci + 1} is diploid")
    t = time.time()
    contigs = []  # (pileup, truth or None for the diploid chromosome)
    all_parts, dip_parts = [], []
    for ci, L in zip(mine, share):
        n_parts = max(1, min(16, L // 4_000_000))
        dip = ci == dip_ci
        with ThreadPoolExecutor(n_parts) as ex:
            parts = list(ex.map(lambda i: Synth(L // n_parts, depth=30, seed=9000 + 100 * ci + i, diploid=dip), range(n_parts)))
        pu = concat_pileups([p.pileup for p in parts], f"chr{ci + 1}")
        contigs.append((pu, None if dip else b"".join(p.hap1 for p in parts)))
        all_parts += parts
        if dip:
            dip_parts = parts
    log(f"pileups generated in {time.time() - t:.1f} s: {sum(pu.n_columns() for pu, _ in contigs) / 1e9:.2f} G columns")
    t = time.time()
    rng = np.random.default_rng(4)
    real = [Synth.yak_assembly(all_parts, k, threads=usable_cpus()) for k in (21, 31)]  # the k-mers of the share's haplotypes
    yaks = [pad_table(y, table_words, rng) for y in real]
    piece = min(dip_parts, key=lambda p: p.pileup.L) if dip_parts else None  # (a generated piece: a >= 1 Mb diploid contig)
    del all_parts, dip_parts
    log(f"k-mer tables: {[len(y.words) for y in real]} words of the haplotypes, padded to {[len(y.words) for y in yaks]} in {time.time() - t:.1f} s")
    free0, total = torch.cuda.mem_get_info()
    t = time.time()
    pol = Polisher(yaks)
    log(f"tables in HBM in {time.time() - t:.1f} s")
    opts = Opts(use_all_reads=True)  # -r
    low = free0
    outs = []
    for rep in range(2):
        res, t_all = [], 0.0
        for pu, truth in contigs:
            c = pol.upload(pu)
            t = time.time()
            b, p = pol.polish_resident(c, opts)
            t_all += time.time() - t
            low = min(low, torch.cuda.mem_get_info()[0])
            assert np.all(p[1:] >= p[:-1]) and int(p[0]) == 0 and int(p[-1]) == pu.L - 1
            res.append(b.tobytes())
            c.free()
        outs.append(res)
        log(f"run {rep}: {sum(share) / t_all / 1e6:.0f} Mbp/s over the share ({t_all * 1e3:.0f} ms of polish calls)")
    log(f"device memory: {total / 2**30:.0f} GiB, high-water mark of this process {(free0 - low) / 2**30:.1f} GiB "
        f"(tables + the longest contig's pileup and scratch)")
    assert outs[0] == outs[1]
    wrong = [pu.name for (pu, truth), b in zip(contigs, outs[0]) if truth is not None and b != truth]
    assert not wrong, wrong
    assert all(pu.ref.tobytes() != b for (pu, truth), b in zip(contigs, outs[0]))
    if piece is not None:
        # -r on a diploid contig of the share against the oracle, stage by stage.  The oracle gets the haplotypes' own
        # words: the padding keys lie above every real key of that k, no lookup can find them (pad_table).
        t = time.time()
        o = Oracle(real)
        o.set_trace(True)
        ob, op = o.polish(piece.pileup, opts)
        t_orc = time.time() - t
        pol.set_trace(True)
        gb, gp = pol.polish(piece.pileup, opts)
        for ps in range(opts.iter_count):
            for st in STAGES:
                a, b = o.trace(ps, st), pol.trace(ps, st)
                assert (a is None) == (b is None), (ps, st)
                if a is not None:
                    assert a.shape == b.shape and np.array_equal(a, b), f"-r, diploid piece: pass {ps} stage {st} differs"
        assert np.array_equal(ob, gb) and np.array_equal(op, gp)
        hete, removed = o.trace(0, "hete.lable"), o.trace(0, "invalid_ids")
        n_hete = int(np.count_nonzero(hete))
        # the same piece WITHOUT -r: reads that disagree with the contig's candidate are dropped before the vote
        # (main.rs:977, 1004-1010), so more reads go — the switch switches
        pol.polish(piece.pileup, Opts())
        removed_default = pol.trace(0, "invalid_ids")
        pol.set_trace(False)
        log(f"diploid piece of {piece.pileup.L} bp under -r == oracle at every stage (oracle {t_orc:.1f} s): {n_hete} marked regions, "
            f"{len(removed)} reads removed by the vote (-r) vs {len(removed_default)} (default)")
        assert piece.pileup.L >= min(1_000_000, int(4_000_000 * scale)) and n_hete > 100 and len(removed) > 0
        assert len(removed_default) > len(removed)
    return sum(share)


def test_one_gpus_share_of_a_t2t_assembly_with_all_reads_kept():
    scale = float(os.environ.get("NP2_T2T_SCALE", "1"))
    run_share(scale, int(1e9 * max(scale, 0.1)))
