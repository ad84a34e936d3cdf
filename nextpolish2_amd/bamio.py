"""Minimal BAM + BAI writer (BGZF on zlib) for synthetic test inputs, and pileup <-> record helpers.

The product reads BAM through csrc/np2_io.cpp; this writer exists so that tests and examples can create
coordinate-sorted, indexed BAM files without samtools/minimap2 (absent here; reference test/hh.sh:8-10)."""
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"
SEQ4 = "=ACMGRSVTWYHKDBN"
_ENC4 = {c: i for i, c in enumerate(SEQ4)}
_ENC4.update({c.lower(): i for i, c in enumerate(SEQ4) if c.isalpha()})


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class _Bgzf:
    def __init__(self, f):
        self.f = f
        self.buf = bytearray()

    def tell(self):
        return (self.f.tell() << 16) | len(self.buf)

    def _flush_block(self, data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(bytes(data)) + co.flush()
        bsize = len(comp) + 25
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
        self.f.write(hdr + comp + struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))

    def write(self, data, atomic=True):
        if atomic and len(self.buf) + len(data) > 60000 and self.buf:
            self.flush()
        self.buf += data
        while len(self.buf) > 64000:
            self._flush_block(self.buf[:60000])
            del self.buf[:60000]

    def flush(self):
        if self.buf:
            self._flush_block(self.buf)
            self.buf = bytearray()

    def close(self):
        self.flush()
        self._flush_block(b"")  # EOF marker


def encode_record(tid, pos, mapq, flag, cigar, seq, name=b"r"):
    """cigar: list of (op_char, len); seq: str/bytes of read bases (may be empty)."""
    if isinstance(seq, str):
        seq = seq.encode()
    name = name + b"\0"
    ref_len = sum(l for op, l in cigar if op in "MDN=X")
    ncig = len(cigar)
    cig = b"".join(struct.pack("<I", (l << 4) | CIGAR_OPS.index(op)) for op, l in cigar)
    codes = [_ENC4.get(chr(c), 15) for c in seq]
    if len(codes) & 1:
        codes.append(0)
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    qual = b"\xff" * len(seq)
    core = struct.pack("<iiBBHHHIiii", tid, pos, len(name), mapq, reg2bin(pos, pos + max(ref_len, 1)), ncig, flag,
                       len(seq), -1, -1, 0)
    body = core + name + cig + packed + qual
    return struct.pack("<I", len(body)) + body, ref_len


def write_bam(path, refs, records):
    """refs: [(name, length)]; records: iterable of dicts(tid,pos,mapq,flag,cigar,seq[,name]) sorted by (tid,pos).

    Writes <path> and <path>.bai."""
    n_ref = len(refs)
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    with open(path, "wb") as f:
        z = _Bgzf(f)
        text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), l) for n, l in refs)
        hdr = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", n_ref)
        for n, l in refs:
            nb = n.encode() + b"\0"
            hdr += struct.pack("<I", len(nb)) + nb + struct.pack("<I", l)
        z.write(hdr, atomic=False)
        z.flush()
        last = (-1, -1)
        for i, r in enumerate(records):
            assert (r["tid"], r["pos"]) >= last, "records must be coordinate sorted"
            last = (r["tid"], r["pos"])
            data, ref_len = encode_record(r["tid"], r["pos"], r.get("mapq", 60), r.get("flag", 0), r["cigar"], r["seq"],
                                          r.get("name", b"r%d" % i))
            if len(z.buf) + len(data) > 60000 and z.buf:
                z.flush()
            beg = z.tell()
            z.write(data, atomic=False)
            end = z.tell()
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
        z.close()
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


def write_bam_raw(path, refs, contigs, level=1, threads=16):
    """Whole-assembly BAM + .bai from already encoded alignment records (Synth.bam_records): contigs = per reference
    sequence, in order, (blob, record offsets, positions, reference lengths).  Records are packed into BGZF blocks of
    at most ~60 kB without splitting one; the blocks are deflated by a thread pool (zlib releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    n_ref = len(refs)
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), l) for n, l in refs)
    hdr = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", n_ref)
    for n, l in refs:
        nb = n.encode() + b"\0"
        hdr += struct.pack("<I", len(nb)) + nb + struct.pack("<I", l)
    blocks = [hdr]            # uncompressed payload of every BGZF block
    where = []                # per contig: (block index, offset inside the block) of every record + the end of the last
    for blob, off, pos, rlen in contigs:
        w, cur, fill = [], bytearray(), 0
        n = len(pos)
        for i in range(n):
            a, b = int(off[i]), int(off[i + 1])
            if fill and fill + (b - a) > 60000:
                blocks.append(bytes(cur))
                cur, fill = bytearray(), 0
            w.append((len(blocks), fill))
            cur += blob[a:b]
            fill += b - a
        w.append((len(blocks), fill))
        blocks.append(bytes(cur))
        where.append(w)

    def deflate(data):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    with ThreadPoolExecutor(max_workers=threads) as pool:
        comp = list(pool.map(deflate, blocks))
    comp.append(deflate(b""))  # EOF marker
    file_off, o = [], 0
    for c in comp:
        file_off.append(o)
        o += len(c)
    with open(path, "wb") as f:
        for c in comp:
            f.write(c)
    voff = lambda bi, within: (file_off[bi] << 16) | within if within < 65536 else None
    with open(path + ".bai", "wb") as f:
        out = b"BAI\1" + struct.pack("<I", n_ref)
        for tid in range(n_ref):
            _, _, pos, rlen = contigs[tid]
            w = where[tid]
            bins, lin = {}, {}
            for i in range(len(pos)):
                beg = voff(*w[i])
                nb, nw = w[i + 1]
                end = voff(nb, nw) if nw else (file_off[nb] << 16)  # (a record ending a block: start of the next one)
                p0, p1 = int(pos[i]), int(pos[i]) + max(int(rlen[i]), 1)
                ch = bins.setdefault(reg2bin(p0, p1), [])
                if ch and ch[-1][1] == beg:
                    ch[-1][1] = end
                else:
                    ch.append([beg, end])
                for win in range(p0 >> 14, ((p1 - 1) >> 14) + 1):
                    lin.setdefault(win, beg)
            out += struct.pack("<I", len(bins))
            for b, ch in sorted(bins.items()):
                out += struct.pack("<II", b, len(ch))
                for beg, end in ch:
                    out += struct.pack("<QQ", beg, end)
            n_intv = (max(lin) + 1) if lin else 0
            out += struct.pack("<I", n_intv)
            prev = 0
            for win in range(n_intv):
                prev = lin.get(win, prev)
                out += struct.pack("<Q", prev)
        f.write(out)


def read_bam(path):
    """-> (refs [(name, length)], records) of a BAM file, records as the dicts write_bam takes.  A plain sequential
    reader for tests (BGZF blocks are concatenated gzip members); the product reads BAM through csrc/np2_io.cpp."""
    import gzip
    with gzip.open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<I", data, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<I", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<I", data, o)
        name = data[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<I", data, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    recs = []
    lut = np.frombuffer(SEQ4.encode(), dtype=np.uint8)
    while o < len(data):
        bs, = struct.unpack_from("<I", data, o)
        tid, pos, l_name, mapq, _bin, ncig, flag, l_seq = struct.unpack_from("<iiBBHHHI", data, o + 4)
        q = o + 36
        name = data[q:q + l_name - 1]
        q += l_name
        cw = struct.unpack_from("<%dI" % ncig, data, q)
        q += 4 * ncig
        nib = np.frombuffer(data, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q)
        codes = np.empty(2 * nib.shape[0], np.uint8)
        codes[0::2] = nib >> 4
        codes[1::2] = nib & 15
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, flag=flag, name=name,
                         cigar=[(CIGAR_OPS[w & 15], w >> 4) for w in cw], seq=lut[codes[:l_seq]].tobytes().decode()))
        o += 4 + bs
    return refs, recs


def pileup_to_records(pileup, tid=0, rng=None, decorate=False):
    """Turn the packed reads (index >= 1) of a Pileup back into BAM-style records (CIGAR + SEQ).

    decorate=True adds soft clips, a few junk columns before/after the 8-match anchors and some records that
    the admission filters must reject, to exercise fill_with_cigar / is_clip / trim / the filters."""
    rng = rng or np.random.default_rng(0)
    code2 = "ACGT-NM"
    ref = pileup.ref.tobytes().decode()
    recs = []
    for r in range(1, pileup.n_reads):
        rd = pileup.reads[r]
        n = int(rd["n_cols"])
        b = pileup.nibbles[int(rd["nib_off"]):int(rd["nib_off"]) + (n + 1) // 2 + 1]
        nib = np.empty(2 * len(b), dtype=np.uint8)
        nib[0::2] = b >> 4
        nib[1::2] = b & 15
        nib = nib[:n]
        ops, seq = [], []
        for c in nib:
            q = c & 7
            if c & 8:
                op = "I"
            elif q == 4:
                op = "D"
            else:
                op = "M"
            if op != "D":
                seq.append(code2[q])
            if ops and ops[-1][0] == op:
                ops[-1][1] += 1
            else:
                ops.append([op, 1])
        pos = int(rd["aln_t_s"])
        flag, mapq = (16 if rng.random() < 0.5 else 0), 60
        if decorate:
            k = int(rng.integers(0, 4))
            if k == 1 and pos >= 12:  # junk mismatching columns before the anchor
                j = int(rng.integers(1, 6))
                junk = "".join("ACGT"[("ACGT".index(ref[pos - j + t].upper()) + 1) % 4] if ref[pos - j + t].upper() in "ACGT" else "A"
                               for t in range(j))
                ops = [["M", j]] + ops if ops[0][0] != "M" else [["M", ops[0][1] + j]] + ops[1:]
                seq = list(junk) + seq
                pos -= j
            if k == 2:  # soft clips of various sizes (some beyond -c 100)
                s1, s2 = int(rng.integers(0, 160)), int(rng.integers(0, 160))
                if s1:
                    ops = [["S", s1]] + ops
                    seq = list("ACGT"[i] for i in rng.integers(0, 4, s1)) + seq
                if s2:
                    ops = ops + [["S", s2]]
                    seq = seq + list("ACGT"[i] for i in rng.integers(0, 4, s2))
            if k == 3:
                ops = [["H", 5]] + ops
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=[(o, l) for o, l in ops], seq="".join(seq)))
        if decorate and rng.random() < 0.08:  # records the filters drop: unmapped / dup / low mapq / supplementary
            bad = dict(recs[-1])
            bad["flag"] = int(rng.choice([4, 0x400, 0x100, 0x800]))
            recs.append(bad)
            low = dict(recs[-2])
            low["mapq"] = 1
            recs.append(low)
    recs.sort(key=lambda x: x["pos"])
    return recs


def records_to_arrays(recs):
    """-> (np2_bamrec array, cigar u32 array, seq4 u8 array, ascii seq blob, ascii offsets) for np2_contig_from_records / the oracle."""
    from .io import BAMREC_DTYPE
    arr = np.zeros(len(recs), dtype=BAMREC_DTYPE)
    cig, seq4, asc, asc_off = [], bytearray(), bytearray(), []
    for i, r in enumerate(recs):
        arr[i]["pos"] = r["pos"]
        arr[i]["flag"] = r.get("flag", 0)
        arr[i]["mapq"] = r.get("mapq", 60)
        arr[i]["n_cigar"] = len(r["cigar"])
        arr[i]["cigar_off"] = len(cig)
        arr[i]["l_seq"] = len(r["seq"])
        arr[i]["seq_off"] = len(seq4)
        asc_off.append(len(asc))
        cig.extend((l << 4) | CIGAR_OPS.index(op) for op, l in r["cigar"])
        codes = [_ENC4.get(c, 15) for c in r["seq"]]
        if len(codes) & 1:
            codes.append(0)
        seq4 += bytes((codes[j] << 4) | codes[j + 1] for j in range(0, len(codes), 2))
        asc += "".join(SEQ4[_ENC4.get(ch, 15)] for ch in r["seq"]).encode()
    seq4 += b"\0" * 16
    return (arr, np.array(cig, dtype=np.uint32), np.frombuffer(bytes(seq4), dtype=np.uint8), bytes(asc),
            np.array(asc_off, dtype=np.uint64))
