"""Fuzz of read extraction from BAM files (run on a GPU box): seeded random multi-reference BAMs written htslib-style —
blocks filled to a random limit whatever the record boundaries, so records and even their fixed fields straddle blocks;
random compression level; decorated records (clips, strands, supplementary, secondary, unmapped mates, low mapq) —
read through np2_contig_from_bam on the DEVICE (BGZF inflate kernel, record walk along the linear index, CIGARs, SEQ in
place) and on the host pool, both compared with each other and with the oracle's front end over the same records.
   python tests/tools/fuzz_bam.py <seed> <files>"""
import os, struct, sys, tempfile, time, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.bamio import encode_record, pileup_to_records, records_to_arrays, reg2bin
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from test_frontend_cpu import same_pileup


def write_bam_straddling(path, refs, records, limit, level):
    """like bamio.write_bam, but a block is cut wherever the buffer reaches `limit` bytes (htslib's bgzf_write)"""
    n_ref = len(refs)
    lin = [dict() for _ in range(n_ref)]
    bins = [dict() for _ in range(n_ref)]
    with open(path, "wb") as f:
        buf = bytearray()

        def flush_block(data):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(bytes(data)) + co.flush()
            if len(comp) + 26 > 65536:  # (does not deflate: stored)
                co = zlib.compressobj(0, zlib.DEFLATED, -15)
                comp = co.compress(bytes(data)) + co.flush()
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))

        def put(data):
            nonlocal buf
            buf += data
            while len(buf) >= limit:
                flush_block(buf[:limit])
                del buf[:limit]
        text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), l) for n, l in refs)
        hdr = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", n_ref)
        for n, l in refs:
            nb = n.encode() + b"\0"
            hdr += struct.pack("<I", len(nb)) + nb + struct.pack("<I", l)
        put(hdr)
        if buf:  # (the header ends its block, as samtools writes it)
            flush_block(buf)
            buf = bytearray()
        for i, r in enumerate(records):
            data, ref_len = encode_record(r["tid"], r["pos"], r.get("mapq", 60), r.get("flag", 0), r["cigar"], r["seq"], r.get("name", b"r%d" % i))
            beg = (f.tell() << 16) | len(buf)
            put(data)
            end = (f.tell() << 16) | len(buf)
            tid = r["tid"]
            if tid >= 0 and not (r.get("flag", 0) & 4):
                b = reg2bin(r["pos"], r["pos"] + max(ref_len, 1))
                ch = bins[tid].setdefault(b, [])
                if ch and ch[-1][1] == beg:
                    ch[-1][1] = end
                else:
                    ch.append([beg, end])
                for w in range(r["pos"] >> 14, ((r["pos"] + max(ref_len, 1) - 1) >> 14) + 1):
                    lin[tid].setdefault(w, beg)
        if buf:
            flush_block(buf)
        flush_block(b"")
    with open(path + ".bai", "wb") as f:
        out = b"BAI\1" + struct.pack("<I", n_ref)
        for tid in range(n_ref):
            out += struct.pack("<I", len(bins[tid]))
            for b, ch in sorted(bins[tid].items()):
                out += struct.pack("<II", b, len(ch))
                for beg, end in ch:
                    out += struct.pack("<QQ", beg, end)
            n_intv = (max(lin[tid]) + 1) if lin[tid] else 0
            out += struct.pack("<I", n_intv)
            prev = 0
            for w in range(n_intv):
                prev = lin[tid].get(w, prev)
                out += struct.pack("<Q", prev)
        f.write(out)


def main():
    global rng
    rng = np.random.default_rng(int(sys.argv[1])); n_files = int(sys.argv[2]); bad = n_ctg = 0
    td = tempfile.mkdtemp()
    pol = Polisher([Synth(2000, seed=3).yak(21)])
    t0 = time.time()
    for case in range(n_files):
        n_ref = int(rng.integers(1, 6))
        syn, recs, refs = [], [], []
        for tid in range(n_ref):
            L = int(rng.choice([2500, 9000, 20000, 50000]))
            seed = int(rng.integers(1, 1 << 30)); rl = float(rng.choice([1500, 4000, 9000]))
            s = Synth(L, depth=int(rng.choice([3, 10, 30])), seed=seed, diploid=bool(rng.integers(0, 2)), read_err_rate=float(rng.choice([0.002, 0.02])),
                      read_len_mean=min(rl, L / 2), read_len_sd=rl / 6, read_len_min=min(1000, L // 4), name=f"c{tid}")
            syn.append(s)
            refs.append((f"c{tid}", s.pileup.L))
            if rng.random() < 0.9:  # (a reference without any record now and then)
                recs += pileup_to_records(s.pileup, tid=tid, rng=np.random.default_rng(seed), decorate=True)
        recs.sort(key=lambda r: (r["tid"], r["pos"]))
        path = os.path.join(td, f"f{case}.bam")
        write_bam_straddling(path, refs, recs, int(rng.choice([300, 4096, 20000, 0xff00])), int(rng.integers(0, 10)))
        fo = np2io.FrontOpts(use_supplementary=bool(rng.integers(0, 2)), min_map_qual=int(rng.choice([0, 1, 30])), min_read_len=int(rng.choice([500, 1000, 2000])),
                             min_map_len=int(rng.choice([200, 500, 1500])), min_map_fra=float(rng.choice([0.2, 0.5, 0.9])), max_clip_len=int(rng.choice([0, 10, 100, 100000])))
        for tid, s in enumerate(syn):
            n_ctg += 1
            rr = [r for r in recs if r["tid"] == tid]
            ref = s.pileup.ref.tobytes()
            try:
                arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
                exp = orc.front_end(ref, arr, cig, asc, asc_off, fo); oerr = None
            except Exception as e:
                oerr = str(e)[:60]
            got = {}
            for mode in ("gpu", "libdeflate"):
                os.environ["NP2_INFLATE"] = mode
                try:
                    c = np2io.contig_from_bam(pol, np2io.Bam(path), f"c{tid}", ref, fo)
                    got[mode] = np2io.export_contig(pol, c, s.pileup.ref)
                    c.free()
                except Np2Error as e:
                    got[mode] = str(e)[:60]
            a, b = got["gpu"], got["libdeflate"]
            if isinstance(a, str) or isinstance(b, str) or oerr:
                if not (isinstance(a, str) and isinstance(b, str) and oerr):
                    bad += 1; print("ERR-MISMATCH", sys.argv[1], case, tid, oerr, a if isinstance(a, str) else "ok", b if isinstance(b, str) else "ok", flush=True)
                continue
            # (byte-wise first; the up to 15 alignment bytes between two reads' streams are nobody's and may differ: then column by column)
            if not (np.array_equal(a.reads, b.reads) and (np.array_equal(a.nibbles, b.nibbles) or same_pileup(a, b))):
                bad += 1
                print("MISMATCH device vs host pool", sys.argv[1], case, tid, "device == oracle:", same_pileup(a, exp), "host pool == oracle:", same_pileup(b, exp),
                      "reads", a.n_reads, b.n_reads, exp.n_reads, "nib bytes", len(a.nibbles), len(b.nibbles), flush=True)
                if os.environ.get("NP2_FUZZ_KEEP"):
                    import shutil; shutil.copy(path, os.environ["NP2_FUZZ_KEEP"] + f"/bad_{sys.argv[1]}_{case}_{tid}.bam")
                continue
            if not same_pileup(a, exp):
                bad += 1; print("MISMATCH vs oracle", sys.argv[1], case, tid, flush=True)
        os.remove(path); os.remove(path + ".bai")
    print(f"bam files {n_files} contigs {n_ctg} bad {bad} time {time.time() - t0:.1f}")


if __name__ == "__main__":
    main()
