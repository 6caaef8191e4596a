"""Seeded synthetic read sets for the hot path (SURVEY.md §8d / BASELINE.md §3).

The reference's own preprocessing (mergereads, extractorfs, translatenucs: out of scope) turns
2x150 bp reads into a DB of ~49-residue protein fragments, which is what the hot path (kmermatcher ->
rescorediagonal -> assembleresults) consumes.  This module produces such a fragment DB directly and
deterministically with numpy so that tests and bench.py need nothing from the reference at run time:

  genome  = concatenated genes (ATG + 300..1500 sense codons + stop, random strand, 50..200 nt spacers)
  reads   = pairs from inserts ~N(320,40) >= 160, 2x150 nt, 0.2 % substitutions
  protein fragments = for each read and each of the 6 frames the stop-free stretch if it is >= 45 codons
                      (a terminal stop is kept as '*', like `translatenucs --add-orf-stop`)
  nucleotide DB     = the reads themselves (PenguiN's nucleotide stage)

DB layout is the MMseqs one (entries "SEQ\\n\\0", keys 0..N-1).
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"   # codon index = 16*b0+4*b1+b2 with A0 C1 G2 T3


def _codon_table():
    return np.frombuffer(_AA.encode(), dtype=np.uint8)


def make_genome(rng, genome_len):
    """nucleotide codes (A0 C1 G2 T3) of a gene-dense genome of about genome_len bases"""
    tab = _codon_table()
    sense = np.nonzero(tab != ord("*"))[0]
    stops = np.nonzero(tab == ord("*"))[0]
    parts, total = [], 0
    while total < genome_len:
        n = int(rng.integers(300, 1501))
        cod = np.concatenate(([14], rng.choice(sense, size=n), [rng.choice(stops)]))      # 14 = ATG
        gene = np.stack([cod // 16, (cod // 4) % 4, cod % 4], axis=1).reshape(-1).astype(np.uint8)
        if rng.random() < 0.5:
            gene = (3 - gene)[::-1]
        spacer = rng.integers(0, 4, size=int(rng.integers(50, 201)), dtype=np.uint8)
        parts += [gene, spacer]
        total += gene.size + spacer.size
    return np.concatenate(parts)


def make_reads(rng, genome, n_pairs, read_len=150, err=0.002):
    """returns uint8 codes [2*n_pairs, read_len] (mate 1 then mate 2 interleaved per pair)"""
    ins = np.maximum(160, rng.normal(320, 40, size=n_pairs).astype(np.int64))
    pos = (rng.random(n_pairs) * (genome.size - ins - 1)).astype(np.int64)
    flip = rng.random(n_pairs) < 0.5
    idx = np.arange(read_len)
    fwd = genome[pos[:, None] + idx[None, :]]                                    # left end, forward strand
    rev = 3 - genome[(pos + ins)[:, None] - 1 - idx[None, :]]                    # right end, reverse complement
    a = np.where(flip[:, None], rev, fwd)
    b = np.where(flip[:, None], fwd, rev)
    reads = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    reads[0::2] = a
    reads[1::2] = b
    mut = rng.random(reads.shape) < err
    reads[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
    return reads


def protein_fragments(reads, min_codons=45):
    """6-frame stop-free stretches >= min_codons; returns (data bytes, off u64, elen u32, key u32).
    Fragments are ordered by (read, frame) like the reference's fragment DB; fully vectorised."""
    flats, lens, keys = [], [], []
    for b in range(0, reads.shape[0], 65536):                 # bound the temporaries
        f, l, k = _fragments_batch(reads[b:b + 65536], min_codons)
        flats.append(f); lens.append(l); keys.append(k + 6 * b)
    flat = np.concatenate(flats); ln = np.concatenate(lens).astype(np.int64); sk = np.concatenate(keys)
    order = np.argsort(sk, kind="stable")
    src_off = np.zeros(ln.size, dtype=np.int64)
    if ln.size:
        src_off[1:] = np.cumsum(ln[:-1])
    elen_sorted = (ln[order] + 2).astype(np.uint32)
    dst_off_sorted = np.zeros(ln.size, dtype=np.int64)
    if ln.size:
        dst_off_sorted[1:] = np.cumsum(elen_sorted[:-1].astype(np.int64))
    dst_off = np.empty_like(dst_off_sorted)
    dst_off[order] = dst_off_sorted
    data = np.zeros(int(elen_sorted.sum()), dtype=np.uint8)
    idx = np.arange(flat.size, dtype=np.int64) + np.repeat(dst_off - src_off, ln)
    data[idx] = flat
    data[dst_off + ln] = 10                                   # '\n' ('\0' is already there)
    key = np.arange(ln.size, dtype=np.uint32)
    return data.tobytes(), dst_off_sorted.astype(np.uint64), elen_sorted, key


def _fragments_batch(reads, min_codons):
    """returns (flat residues of all fragments, fragment lengths, sort keys row*6+frame), grouped by frame"""
    tab = _codon_table()
    n, L = reads.shape
    rc = (3 - reads)[:, ::-1]
    flats, lens, keys = [], [], []
    frame = 0
    for strand in (reads, rc):
        for f in range(3):
            nc = (L - f) // 3
            c = strand[:, f:f + 3 * nc].reshape(n, nc, 3).astype(np.int64)
            aa = tab[c[:, :, 0] * 16 + c[:, :, 1] * 4 + c[:, :, 2]]             # [n, nc]
            stop = aa == ord("*")
            slack = nc - min_codons
            # a qualifying stretch starts after the last stop of the first `slack` codons and ends at the first
            # stop of the last `slack` codons (kept as '*'), with no stop in between
            if slack > 0:
                head, tail = stop[:, :slack], stop[:, nc - slack:]
                start = np.where(head.any(1), slack - np.argmax(head[:, ::-1], axis=1), 0)
                end = np.where(tail.any(1), nc - slack + np.argmax(tail, axis=1) + 1, nc)
                endx = np.where(tail.any(1), end - 1, end)                       # exclusive end of the stop-free part
            else:
                start = np.zeros(n, np.int64); end = np.full(n, nc); endx = end
            csum = np.concatenate([np.zeros((n, 1), np.int64), np.cumsum(stop, axis=1)], axis=1)
            inner = csum[np.arange(n), endx] - csum[np.arange(n), start]
            ok = (inner == 0) & ((endx - start) >= min_codons)
            rows = np.nonzero(ok)[0]
            col = np.arange(nc)[None, :]
            m = (col >= start[rows, None]) & (col < end[rows, None])
            flats.append(aa[rows][m]); lens.append((end[rows] - start[rows]).astype(np.int64)); keys.append(rows.astype(np.int64) * 6 + frame)
            frame += 1
    return np.concatenate(flats), np.concatenate(lens), np.concatenate(keys)


def pack_db(seqs):
    """list of uint8 arrays -> MMseqs entries "SEQ\\n\\0" with keys 0..N-1"""
    n = len(seqs)
    elen = np.fromiter((s.size + 2 for s in seqs), dtype=np.uint32, count=n)
    off = np.zeros(n, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum(elen[:-1], dtype=np.uint64)
    data = np.zeros(int(elen.sum()), dtype=np.uint8)
    for s, o in zip(seqs, off):
        o = int(o)
        data[o:o + s.size] = s
        data[o + s.size] = 10
    key = np.arange(n, dtype=np.uint32)
    return data.tobytes(), off, elen, key


def protein_fragment_db(n_pairs, genome_len=None, seed=1, coverage=20.0):
    """Config-C2-style protein fragment DB. genome_len defaults to 2*150*n_pairs/coverage."""
    rng = np.random.default_rng(seed)
    if genome_len is None:
        genome_len = max(20000, int(300 * n_pairs / coverage))
    genome = make_genome(rng, genome_len)
    reads = make_reads(rng, genome, n_pairs)
    return protein_fragments(reads)


def nucleotide_read_db(n_pairs, genome_len=None, seed=1, coverage=20.0):
    rng = np.random.default_rng(seed)
    if genome_len is None:
        genome_len = max(20000, int(300 * n_pairs / coverage))
    genome = make_genome(rng, genome_len)
    reads = make_reads(rng, genome, n_pairs)
    return pack_db([_BASES[r] for r in reads])


def plant_inverted_repeats(rng, genome, n):
    """writes n inverted repeats (arm 40-150 nt + loop 0-60 nt + reverse complement of the arm) over the genome: loci where reads share
    k-mers on BOTH strands — the (rep, target, diagonal) strand ties of kmermatcher's sort #2 (tests/golden/make_strand_ties.py)"""
    for _ in range(n):
        arm, loop = int(rng.integers(40, 151)), int(rng.integers(0, 61))
        p0 = int(rng.integers(1000, genome.size - 1000))
        genome[p0 + arm + loop:p0 + 2 * arm + loop] = (3 - genome[p0:p0 + arm])[::-1]
    return genome


def nucleotide_hairpin_reads(n_pairs, n_hairpins, seed=1, coverage=60.0):
    """read codes [2 * n_pairs, 150] of ONE gene-dense genome with planted inverted repeats; returns (reads, genome length)"""
    rng = np.random.default_rng(seed)
    genome = plant_inverted_repeats(rng, make_genome(rng, max(20000, int(300 * n_pairs / coverage))), n_hairpins)
    return make_reads(rng, genome, n_pairs), int(genome.size)


def fixed_length_db(reads):
    """(data, off, elen, key) of equally long reads (codes A0 C1 G2 T3) without a Python loop"""
    n, L = reads.shape
    ent = np.empty((n, L + 2), dtype=np.uint8)
    ent[:, :L] = _BASES[reads]; ent[:, L] = 10; ent[:, L + 1] = 0
    return ent.tobytes(), np.arange(n, dtype=np.uint64) * (L + 2), np.full(n, L + 2, dtype=np.uint32), np.arange(n, dtype=np.uint32)


def orf_twin_dbs(n_pairs, genome_len=None, seed=1, coverage=20.0, min_codons=45):
    """PenguiN's guided stage works on nucleotide ORFs and their translations under the same keys (extractorfs +
    translatenucs --add-orf-stop).  Returns (nucl_db, aa_db), each (data, off, elen, key): for every read and frame the
    stop-free stretch of >= min_codons codons (a terminal stop codon is kept: '*' in the protein twin), the nucleotide
    entry being exactly the 3*len codons the protein entry translates.  Simple per-read loop (test sizes)."""
    rng = np.random.default_rng(seed)
    if genome_len is None:
        genome_len = max(20000, int(300 * n_pairs / coverage))
    genome = make_genome(rng, genome_len)
    reads = make_reads(rng, genome, n_pairs)
    tab = _codon_table()
    nucl, aa = [], []
    for r in reads:
        for strand in (r, (3 - r)[::-1]):
            for f in range(3):
                nc = (strand.size - f) // 3
                c = strand[f:f + 3 * nc].reshape(nc, 3).astype(np.int64)
                prot = tab[c[:, 0] * 16 + c[:, 1] * 4 + c[:, 2]]
                stops = np.nonzero(prot == ord("*"))[0]
                # stop-free stretches (a stretch may end WITH its stop codon)
                begin = 0
                for e in list(stops) + [nc]:
                    end = min(e + 1, nc) if e < nc else nc          # include the stop codon when there is one
                    if (e - begin) >= min_codons:
                        aa.append(prot[begin:end].copy()); nucl.append(_BASES[strand[f + 3 * begin:f + 3 * end]])
                    begin = e + 1
    return pack_db(nucl), pack_db(aa)


def write_db(path, data, off, elen, key, dbtype=0):
    with open(path, "wb") as f:
        f.write(data)
    with open(path + ".index", "wb") as f:
        for k, o, l in zip(key, off, elen):
            f.write(b"%d\t%d\t%d\n" % (int(k), int(o), int(l)))
    with open(path + ".dbtype", "wb") as f:
        f.write(int(dbtype).to_bytes(4, "little"))
