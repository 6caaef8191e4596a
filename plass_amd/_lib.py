"""ctypes binding of include/plasship.h (no torch types cross the boundary)."""
import ctypes as C
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    """the in-tree C-ABI library; PLASSHIP_LIB names another build of it (A/B runs of kernel variants)"""
    return os.environ.get("PLASSHIP_LIB") or os.path.join(_HERE, "libplasship.so")


class PlasshipError(RuntimeError):
    pass


class _KmermatchParams(C.Structure):
    _fields_ = [("kmer_size", C.c_int32), ("alphabet_size", C.c_int32), ("kmers_per_seq", C.c_int32),
                ("kmers_per_seq_scale", C.c_float), ("hash_shift", C.c_int32), ("include_only_extendable", C.c_int32),
                ("ignore_multi_kmer", C.c_int32), ("cov_mode", C.c_int32), ("cov_thr", C.c_float)]


class KmermatchStats(C.Structure):
    _fields_ = [("n_kmer_records", C.c_uint64), ("n_grouped", C.c_uint64), ("n_candidates", C.c_uint64),
                ("record_bytes", C.c_uint32), ("ms_extract", C.c_float), ("ms_sort1", C.c_float),
                ("ms_group", C.c_float), ("ms_sort2", C.c_float), ("ms_reduce", C.c_float), ("ms_extract_kernel", C.c_float),
                ("residues", C.c_uint64), ("ms_extract_short_kernel", C.c_float), ("ms_extract_wave_kernel", C.c_float),
                ("short_residues", C.c_uint64), ("short_records", C.c_uint64), ("wave_residues", C.c_uint64), ("wave_records", C.c_uint64),
                ("ms_part_scatter", C.c_float), ("n_part_scatter", C.c_int32), ("n_scratch_sequences", C.c_uint32), ("n_restarts", C.c_uint32),
                ("n_cached_sequences", C.c_uint32), ("reserved0", C.c_uint32)]


class _RescoreParams(C.Structure):
    _fields_ = [("rescore_mode", C.c_int32), ("eval_thr", C.c_double), ("seq_id_thr", C.c_float), ("cov_mode", C.c_int32),
                ("cov_thr", C.c_float), ("min_aln_len", C.c_int32), ("seq_id_mode", C.c_int32), ("add_backtrace", C.c_int32),
                ("include_identity", C.c_int32)]


class RescoreStats(C.Structure):
    _fields_ = [("n_scored", C.c_uint64), ("n_accepted", C.c_uint64), ("overlap_residues", C.c_uint64), ("ms_kernel", C.c_float)]


class _AssembleParams(C.Structure):
    _fields_ = [("seq_id_thr", C.c_float), ("max_seq_len", C.c_uint64), ("keep_target", C.c_int32), ("rescore_mode", C.c_int32)]


class AssembleStats(C.Structure):
    _fields_ = [("n_extended", C.c_uint64), ("n_rescored", C.c_uint64), ("out_residues", C.c_uint64), ("ms_kernel", C.c_float),
                ("ms_assemble_kernel", C.c_float), ("n_alignments", C.c_uint64), ("rescored_residues", C.c_uint64),
                ("ms_tier_kernel", C.c_float * 3), ("tier_alignments", C.c_uint64 * 3), ("tier_query_residues", C.c_uint64 * 3),
                ("tier_rescored_residues", C.c_uint64 * 3), ("db_appended_bytes", C.c_uint64), ("db_copied_bytes", C.c_uint64)]


class _Aln2NuclParams(C.Structure):
    _fields_ = [("gap_open", C.c_int32), ("gap_extend", C.c_int32)]


class Aln2NuclStats(C.Structure):
    _fields_ = [("n_alignments", C.c_uint64), ("ms_kernel", C.c_float)]


class FindStartStats(C.Structure):
    _fields_ = [("n_alignments", C.c_uint64), ("out_residues", C.c_uint64), ("ms_kernel", C.c_float)]


class _CyclecheckParams(C.Structure):
    _fields_ = [("max_seq_len", C.c_uint64), ("chop_cycle", C.c_int32)]


class CyclecheckStats(C.Structure):
    _fields_ = [("n_cyclic", C.c_uint64), ("n_wave_small", C.c_uint64), ("n_wave_large", C.c_uint64), ("n_block", C.c_uint64), ("ms_kernel", C.c_float), ("n_known", C.c_uint64)]


class _OrfParams(C.Structure):
    _fields_ = [("min_length", C.c_int32), ("max_length", C.c_int32), ("max_gaps", C.c_int32), ("contig_start_mode", C.c_int32),
                ("contig_end_mode", C.c_int32), ("orf_start_mode", C.c_int32), ("forward_frames", C.c_int32), ("reverse_frames", C.c_int32),
                ("translation_table", C.c_int32), ("translate", C.c_int32), ("use_all_table_starts", C.c_int32), ("max_seq_len", C.c_uint64)]


class _TranslateParams(C.Structure):
    _fields_ = [("translation_table", C.c_int32), ("add_orf_stop", C.c_int32), ("max_seq_len", C.c_uint64)]


class OrfStats(C.Structure):
    _fields_ = [("n_out", C.c_uint64), ("in_residues", C.c_uint64), ("out_residues", C.c_uint64), ("ms_kernel", C.c_float)]


class _SynthParams(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("seed", C.c_uint64), ("n_genomes", C.c_uint32), ("genome_min_len", C.c_uint64),
                ("genome_max_len", C.c_uint64), ("abundance_sigma", C.c_float), ("insert_mean", C.c_float), ("insert_sd", C.c_float),
                ("insert_min", C.c_uint32), ("read_len", C.c_uint32), ("error_rate", C.c_float)]


class SynthStats(C.Structure):
    _fields_ = [("genome_bases", C.c_uint64), ("n_genes", C.c_uint64), ("mean_coverage", C.c_double), ("max_coverage", C.c_double),
                ("ms_kernel", C.c_float)]


class AlnRecord(C.Structure):
    _fields_ = [("query_key", C.c_uint32), ("target_key", C.c_uint32), ("bit_score", C.c_int32), ("raw_score", C.c_int32),
                ("seq_id", C.c_float), ("q_start", C.c_int32), ("q_end", C.c_int32), ("q_len", C.c_int32),
                ("db_start", C.c_int32), ("db_end", C.c_int32), ("db_len", C.c_int32), ("aln_len", C.c_int32), ("reversed", C.c_int32)]


# every symbol include/plasship.h declares: (name, restype, argtypes)
P = C.c_void_p
SYMBOLS = [
    ("plasship_last_error", C.c_char_p, []),
    ("plasship_version", C.c_char_p, []),
    ("plasship_ctx_create", C.c_int, [C.c_int, C.POINTER(P)]),
    ("plasship_ctx_destroy", None, [P]),
    ("plasship_ctx_sync", C.c_int, [P]),
    ("plasship_ctx_reserve_async", C.c_int, [P]),
    ("plasship_ctx_stream", P, [P]),
    ("plasship_host_syncs", C.c_ulonglong, []),
    ("plasship_ctx_debug_fail_collective", C.c_int, [P, C.c_int]),
    ("plasship_ctx_set_comm", C.c_int, [P, P]),
    ("plasship_ctx_copy_d2d", C.c_int, [P, P, P, C.c_uint64]),
    ("plasship_seqdb_upload", C.c_int, [P, C.c_char_p, C.c_size_t, P, P, P, C.c_size_t, C.c_int, C.POINTER(P)]),
    ("plasship_seqdb_read", C.c_int, [P, C.c_char_p, C.POINTER(P)]),
    ("plasship_seqdb_write", C.c_int, [P, P, C.c_char_p]),
    ("plasship_seqdb_info", C.c_int, [P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    ("plasship_seqdb_digest", C.c_int, [P, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("plasship_seqdb_download", C.c_int, [P, P, P, P, P, P]),
    ("plasship_seqdb_free", None, [P, P]),
    ("plasship_kmermatch", C.c_int, [P, P, C.POINTER(_KmermatchParams), C.POINTER(P), C.POINTER(KmermatchStats)]),
    ("plasship_cands_write", C.c_int, [P, P, P, C.c_char_p]),
    ("plasship_cands_read", C.c_int, [P, P, P, C.c_char_p, C.POINTER(P)]),
    ("plasship_cands_count", C.c_int, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("plasship_cands_download", C.c_int, [P, P, P, P, P, P, P, P]),
    ("plasship_cands_free", None, [P, P]),
    ("plasship_rescore", C.c_int, [P, P, P, P, C.POINTER(_RescoreParams), C.POINTER(P), C.POINTER(RescoreStats)]),
    ("plasship_alns_write", C.c_int, [P, P, C.c_char_p]),
    ("plasship_alns_read", C.c_int, [P, P, C.c_char_p, C.POINTER(P)]),
    ("plasship_alns_count", C.c_int, [P, C.POINTER(C.c_uint64)]),
    ("plasship_alns_download", C.c_int, [P, P, P]),
    ("plasship_alns_free", None, [P, P]),
    ("plasship_assemble", C.c_int, [P, P, P, C.POINTER(_AssembleParams), C.POINTER(P), C.POINTER(AssembleStats)]),
    ("plasship_guided_assemble", C.c_int, [P, P, P, P, C.POINTER(_AssembleParams), C.POINTER(P), C.POINTER(P), C.POINTER(AssembleStats)]),
    ("plasship_aln2nucl", C.c_int, [P, P, P, P, P, P, C.POINTER(_Aln2NuclParams), C.POINTER(P), C.POINTER(Aln2NuclStats)]),
    ("plasship_find_assembly_start", C.c_int, [P, P, P, C.POINTER(P), C.POINTER(FindStartStats)]),
    ("plasship_cyclecheck", C.c_int, [P, P, C.POINTER(_CyclecheckParams), C.POINTER(P), C.POINTER(P), C.POINTER(CyclecheckStats)]),
    ("plasship_extract_orfs", C.c_int, [P, P, C.POINTER(_OrfParams), C.POINTER(P), C.POINTER(P), C.POINTER(OrfStats)]),
    ("plasship_translate_nucs", C.c_int, [P, P, P, C.POINTER(_TranslateParams), C.POINTER(P), C.POINTER(OrfStats)]),
    ("plasship_seqdb_concat", C.c_int, [P, P, P, C.POINTER(P)]),
    ("plasship_seqdb_concat_keys", C.c_int, [P, P, P, C.c_int, C.POINTER(P)]),
    ("plasship_orfhdr_concat", C.c_int, [P, P, P, C.POINTER(P)]),
    ("plasship_orfhdr_read", C.c_int, [P, C.c_char_p, C.POINTER(P)]),
    ("plasship_orfhdr_write", C.c_int, [P, P, C.c_char_p]),
    ("plasship_orfhdr_count", C.c_int, [P, C.POINTER(C.c_size_t)]),
    ("plasship_orfhdr_free", None, [P, P]),
]
# include/plasship_rccl.h (native RCCL communicator of a sharded run)
RCCL_SYMBOLS = [
    ("plasship_rccl_get_unique_id", C.c_int, [P]),
    ("plasship_rccl_comm_create", C.c_int, [P, C.c_int, C.c_int, P, C.POINTER(P)]),
    ("plasship_rccl_comm_stats", C.c_int, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    ("plasship_rccl_comm_destroy", None, [P, P]),
]
# include/plasship_synth.h (measurement infrastructure: synthetic read sets generated on the GPU)
SYNTH_SYMBOLS = [
    ("plasship_synth_read_pairs", C.c_int, [P, C.POINTER(_SynthParams), C.POINTER(P), C.POINTER(SynthStats)]),
]

_lib = None


def load_library():
    """dlopen the in-tree HIP library and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise PlasshipError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(plass_amd has no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS + SYNTH_SYMBOLS + RCCL_SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().plasship_last_error()
        raise PlasshipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


@dataclass
class KmermatchParams:
    """kmermatcher flags (defaults = plass assemble, src/workflow/Assembler.cpp:10-27)."""
    k: int = 14
    alph_size: int = 13
    kmer_per_seq: int = 60
    kmer_per_seq_scale: float = 0.0
    hash_shift: int = 67
    include_only_extendable: bool = False
    ignore_multi_kmer: bool = True
    cov_mode: int = 0
    c: float = 0.0

    def _c(self):
        return _KmermatchParams(self.k, self.alph_size, self.kmer_per_seq, self.kmer_per_seq_scale, self.hash_shift,
                                int(self.include_only_extendable), int(self.ignore_multi_kmer), self.cov_mode, self.c)


@dataclass
class RescoreParams:
    rescore_mode: int = 3
    e: float = 1e-5
    min_seq_id: float = 0.9
    cov_mode: int = 0
    c: float = 0.0
    min_aln_len: int = 0
    seq_id_mode: int = 0
    a: bool = False
    add_self_matches: bool = False

    def _c(self):
        return _RescoreParams(self.rescore_mode, self.e, self.min_seq_id, self.cov_mode, self.c, self.min_aln_len,
                              self.seq_id_mode, int(self.a), int(self.add_self_matches))


@dataclass
class AssembleParams:
    min_seq_id: float = 0.9
    max_seq_len: int = 65535
    keep_target: bool = True
    rescore_mode: int = 3

    def _c(self):
        return _AssembleParams(self.min_seq_id, self.max_seq_len, int(self.keep_target), self.rescore_mode)


@dataclass
class OrfParams:
    """extractorfs flags; defaults = the module's own (mm/commons/Parameters.cpp), frames as the reference's "1,2,3" lists"""
    min_length: int = 30
    max_length: int = 32734
    max_gaps: int = 2147483647
    contig_start_mode: int = 2
    contig_end_mode: int = 2
    orf_start_mode: int = 1
    forward_frames: str = "1,2,3"
    reverse_frames: str = "1,2,3"
    translation_table: int = 1
    translate: bool = False
    use_all_table_starts: bool = False
    max_seq_len: int = 65535

    @staticmethod
    def _frames(v):
        m = 0
        for f in str(v).split(","):
            f = f.strip()
            if f:
                m |= 1 << (int(f) - 1)
        return m

    def _c(self):
        return _OrfParams(self.min_length, self.max_length, self.max_gaps, self.contig_start_mode, self.contig_end_mode, self.orf_start_mode,
                          self._frames(self.forward_frames), self._frames(self.reverse_frames), self.translation_table, int(self.translate),
                          int(self.use_all_table_starts), self.max_seq_len)


# the two extractorfs passes of `plass assemble` (src/workflow/Assembler.cpp:116-130; data/assemble.sh:41-63)
PLASS_ORFS_LONG = dict(min_length=45, max_length=32734, max_gaps=0, contig_start_mode=2, contig_end_mode=2, orf_start_mode=0)
PLASS_ORFS_START = dict(min_length=20, max_length=45, max_gaps=0, contig_start_mode=1, contig_end_mode=0, orf_start_mode=0)


@dataclass
class SynthParams:
    """synthetic community + read pairs (include/plasship_synth.h)"""
    n_pairs: int = 500000
    seed: int = 1
    n_genomes: int = 1
    genome_min_len: int = 7500000
    genome_max_len: int = 7500000
    abundance_sigma: float = 0.0
    insert_mean: float = 320.0
    insert_sd: float = 40.0
    insert_min: int = 160
    read_len: int = 150
    error_rate: float = 0.002

    def _c(self):
        return _SynthParams(self.n_pairs, self.seed, self.n_genomes, self.genome_min_len, self.genome_max_len, self.abundance_sigma,
                            self.insert_mean, self.insert_sd, self.insert_min, self.read_len, self.error_rate)


class Context:
    """One GPU (one process per GPU; LOCAL_RANK picks the device when ordinal < 0)."""

    def __init__(self, device=-1):
        self.lib = load_library()
        self.h = P()
        _check(self.lib.plasship_ctx_create(device, C.byref(self.h)), "plasship_ctx_create")

    def close(self):
        if self.h:
            self.lib.plasship_ctx_destroy(self.h)
            self.h = P()

    def sync(self):
        _check(self.lib.plasship_ctx_sync(self.h), "plasship_ctx_sync")

    def host_syncs(self):
        """host waits for a stream since the library was loaded (diagnostic; bench.py reports the count per iteration)"""
        return int(self.lib.plasship_host_syncs())

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- modules -------------------------------------------------------------------------------
    def read_seqdb(self, path):
        h = P()
        _check(self.lib.plasship_seqdb_read(self.h, os.fsencode(path), C.byref(h)), "plasship_seqdb_read")
        return SeqDB(self, h)

    def upload_seqdb(self, data, off, elen, key, dbtype=0):
        import numpy as np
        data = bytes(data)
        off = np.ascontiguousarray(off, dtype=np.uint64); elen = np.ascontiguousarray(elen, dtype=np.uint32)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        h = P()
        _check(self.lib.plasship_seqdb_upload(self.h, data, len(data), off.ctypes.data, elen.ctypes.data, key.ctypes.data,
                                              len(key), dbtype, C.byref(h)), "plasship_seqdb_upload")
        return SeqDB(self, h)

    def kmermatcher(self, db, par=None):
        par = par or KmermatchParams()
        h = P(); st = KmermatchStats(); cp = par._c()
        _check(self.lib.plasship_kmermatch(self.h, db.h, C.byref(cp), C.byref(h), C.byref(st)), "plasship_kmermatch")
        return Candidates(self, h, db, db), st

    def read_prefdb(self, qdb, tdb, path):
        h = P()
        _check(self.lib.plasship_cands_read(self.h, qdb.h, tdb.h, os.fsencode(path), C.byref(h)), "plasship_cands_read")
        return Candidates(self, h, qdb, tdb)

    def rescorediagonal(self, qdb, tdb, cands, par=None):
        par = par or RescoreParams()
        h = P(); st = RescoreStats(); cp = par._c()
        _check(self.lib.plasship_rescore(self.h, qdb.h, tdb.h, cands.h, C.byref(cp), C.byref(h), C.byref(st)), "plasship_rescore")
        return Alignments(self, h, qdb, tdb), st

    def read_alndb(self, db, path):
        h = P()
        _check(self.lib.plasship_alns_read(self.h, db.h, os.fsencode(path), C.byref(h)), "plasship_alns_read")
        return Alignments(self, h, db, db)

    def assembleresults(self, db, alns, par=None):
        par = par or AssembleParams()
        h = P(); st = AssembleStats(); cp = par._c()
        _check(self.lib.plasship_assemble(self.h, db.h, alns.h, C.byref(cp), C.byref(h), C.byref(st)), "plasship_assemble")
        return SeqDB(self, h), st

    def guidedassembleresults(self, nucl_db, aa_db, alns, par=None):
        """(nuclDB, aaDB, nucleotide-level alignments) -> (nuclDB', aaDB'); reference module guidedassembleresults"""
        par = par or AssembleParams(min_seq_id=0.99, max_seq_len=200000)
        hn = P(); ha = P(); st = AssembleStats(); cp = par._c()
        _check(self.lib.plasship_guided_assemble(self.h, nucl_db.h, aa_db.h, alns.h, C.byref(cp), C.byref(hn), C.byref(ha), C.byref(st)), "plasship_guided_assemble")
        return SeqDB(self, hn), SeqDB(self, ha), st

    def proteinaln2nucl(self, nucl_db, aa_db, alns, gap_open=5, gap_extend=2):
        """protein alignments (with backtrace) of aa_db onto the nucleotide twins in nucl_db; reference module proteinaln2nucl"""
        h = P(); st = Aln2NuclStats(); cp = _Aln2NuclParams(gap_open, gap_extend)
        _check(self.lib.plasship_aln2nucl(self.h, nucl_db.h, nucl_db.h, aa_db.h, aa_db.h, alns.h, C.byref(cp), C.byref(h), C.byref(st)), "plasship_aln2nucl")
        return Alignments(self, h, nucl_db, nucl_db), st

    def findassemblystart(self, db, alns):
        """(protein DB, its alignments) -> DB with the consensus "*M" starts cut in; reference module findassemblystart,
        run once inside iteration 0 of `plass assemble` (data/assemble.sh:110-141)"""
        h = P(); st = FindStartStats()
        _check(self.lib.plasship_find_assembly_start(self.h, db.h, alns.h, C.byref(h), C.byref(st)), "plasship_find_assembly_start")
        return SeqDB(self, h), st

    def cyclecheck(self, db, max_seq_len=200000, chop_cycle=False, with_rest=False):
        """nucleotide DB -> DB of the circular / terminally redundant contigs (and, with_rest, the DB of all others);
        reference module cyclecheck, run after every nuclassembleresults of the penguin workflows"""
        hc = P(); hr = P(); st = CyclecheckStats(); cp = _CyclecheckParams(max_seq_len, int(chop_cycle))
        _check(self.lib.plasship_cyclecheck(self.h, db.h, C.byref(cp), C.byref(hc), C.byref(hr) if with_rest else None, C.byref(st)), "plasship_cyclecheck")
        return (SeqDB(self, hc), SeqDB(self, hr), st) if with_rest else (SeqDB(self, hc), st)

    # ---- row N2: the once-per-run preprocessing (data/assemble.sh:41-77) ------------------------------------------
    def extractorfs(self, reads, par=None):
        """nucleotide DB -> (ORF DB, its header DB); reference module extractorfs"""
        par = par or OrfParams()
        h = P(); hh = P(); st = OrfStats(); cp = par._c()
        _check(self.lib.plasship_extract_orfs(self.h, reads.h, C.byref(cp), C.byref(h), C.byref(hh), C.byref(st)), "plasship_extract_orfs")
        return SeqDB(self, h), OrfHeaders(self, hh), st

    def translatenucs(self, orfs, hdr=None, add_orf_stop=False, max_seq_len=65535):
        """nucleotide ORF DB (+ header DB for --add-orf-stop) -> protein DB; reference module translatenucs"""
        h = P(); st = OrfStats(); cp = _TranslateParams(1, int(add_orf_stop), max_seq_len)
        _check(self.lib.plasship_translate_nucs(self.h, orfs.h, hdr.h if hdr is not None else None, C.byref(cp), C.byref(h), C.byref(st)), "plasship_translate_nucs")
        return SeqDB(self, h), st

    def concatdbs(self, a, b, preserve_keys=False):
        """reference module concatdbs (keys of A kept, B renumbered behind them; preserve_keys: B's kept as well — the union); sequence DBs or header DBs"""
        h = P()
        if preserve_keys:
            _check(self.lib.plasship_seqdb_concat_keys(self.h, a.h, b.h, 1, C.byref(h)), "plasship_seqdb_concat_keys")
            return SeqDB(self, h)
        if isinstance(a, OrfHeaders):
            _check(self.lib.plasship_orfhdr_concat(self.h, a.h, b.h, C.byref(h)), "plasship_orfhdr_concat")
            return OrfHeaders(self, h)
        _check(self.lib.plasship_seqdb_concat(self.h, a.h, b.h, C.byref(h)), "plasship_seqdb_concat")
        return SeqDB(self, h)

    def read_orfhdr(self, path):
        h = P()
        _check(self.lib.plasship_orfhdr_read(self.h, os.fsencode(path), C.byref(h)), "plasship_orfhdr_read")
        return OrfHeaders(self, h)

    def plass_fragments(self, reads, keep=False):
        """the whole preprocessing chain of `plass assemble` on a read DB: two extractorfs passes, translatenucs --add-orf-stop,
        concatdbs -> aa_6f_start_long, the DB iteration 0 starts from (data/assemble.sh:41-77)"""
        o_long, h_long, _ = self.extractorfs(reads, OrfParams(**PLASS_ORFS_LONG))
        aa_long, _ = self.translatenucs(o_long, h_long, add_orf_stop=True)
        o_long.free(); h_long.free()
        o_start, h_start, _ = self.extractorfs(reads, OrfParams(**PLASS_ORFS_START))
        aa_start, _ = self.translatenucs(o_start, h_start, add_orf_stop=True)
        o_start.free(); h_start.free()
        out = self.concatdbs(aa_long, aa_start)
        aa_long.free(); aa_start.free()
        return out

    def penguin_guided_inputs(self, reads):
        """the preprocessing of `penguin guided_nuclassemble` on a read DB (data/guidedNuclAssemble.sh:44-75): two extractorfs passes,
        concatdbs of the ORFs and of their headers, translatenucs --add-orf-stop -> (nucl_6f_start_long, aa_6f_start_long)"""
        o_long, h_long, _ = self.extractorfs(reads, OrfParams(**PLASS_ORFS_LONG))
        o_start, h_start, _ = self.extractorfs(reads, OrfParams(**PLASS_ORFS_START))
        nucl = self.concatdbs(o_long, o_start)
        hdr = self.concatdbs(h_long, h_start)
        for x in (o_long, o_start, h_long, h_start):
            x.free()
        aa, _ = self.translatenucs(nucl, hdr, add_orf_stop=True)
        hdr.free()
        return nucl, aa

    def synth_read_pairs(self, par):
        """synthetic read pairs generated in HBM (include/plasship_synth.h) -> nucleotide read DB"""
        h = P(); st = SynthStats(); cp = par._c()
        _check(self.lib.plasship_synth_read_pairs(self.h, C.byref(cp), C.byref(h), C.byref(st)), "plasship_synth_read_pairs")
        return SeqDB(self, h), st

    # the library picks the variant from the DB type; this name mirrors the reference module for nucleotide DBs
    def nuclassembleresults(self, db, alns, par=None):
        if db.info()["dbtype"] != 1:
            raise ValueError("nuclassembleresults needs a nucleotide sequence DB")
        return self.assembleresults(db, alns, par or AssembleParams(min_seq_id=0.99, max_seq_len=200000))


class SeqDB:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def info(self):
        n = C.c_size_t(); res = C.c_uint64(); mx = C.c_uint32(); ty = C.c_int(); nb = C.c_uint64()
        _check(self.ctx.lib.plasship_seqdb_info(self.h, C.byref(n), C.byref(res), C.byref(mx), C.byref(ty), C.byref(nb)), "plasship_seqdb_info")
        return dict(n=n.value, residues=res.value, max_entry_len=mx.value, dbtype=ty.value, data_bytes=nb.value)

    def write(self, path):
        _check(self.ctx.lib.plasship_seqdb_write(self.ctx.h, self.h, os.fsencode(path)), "plasship_seqdb_write")

    def digest(self):
        """(digest as 16 hex digits, entry bytes): order-independent digest of the resident DB (include/plasship.h: plasship_seqdb_digest)"""
        d = C.c_uint64(); b = C.c_uint64()
        _check(self.ctx.lib.plasship_seqdb_digest(self.ctx.h, self.h, C.byref(d), C.byref(b)), "plasship_seqdb_digest")
        return "%016x" % d.value, b.value

    def download(self):
        import numpy as np
        i = self.info()
        data = C.create_string_buffer(max(i["data_bytes"], 1))
        off = np.zeros(i["n"], dtype=np.uint64); elen = np.zeros(i["n"], dtype=np.uint32); key = np.zeros(i["n"], dtype=np.uint32)
        _check(self.ctx.lib.plasship_seqdb_download(self.ctx.h, self.h, data, off.ctypes.data, elen.ctypes.data, key.ctypes.data), "plasship_seqdb_download")
        return data.raw[:i["data_bytes"]], off, elen, key

    def free(self):
        if self.h:
            self.ctx.lib.plasship_seqdb_free(self.ctx.h, self.h); self.h = P()


class OrfHeaders:
    """header DB of an ORF DB (<db>_h), device resident"""
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def count(self):
        n = C.c_size_t()
        _check(self.ctx.lib.plasship_orfhdr_count(self.h, C.byref(n)), "plasship_orfhdr_count")
        return n.value

    def write(self, path):
        _check(self.ctx.lib.plasship_orfhdr_write(self.ctx.h, self.h, os.fsencode(path)), "plasship_orfhdr_write")

    def free(self):
        if self.h:
            self.ctx.lib.plasship_orfhdr_free(self.ctx.h, self.h); self.h = P()


class Candidates:
    def __init__(self, ctx, h, qdb, tdb):
        self.ctx, self.h, self.qdb, self.tdb = ctx, h, qdb, tdb

    def count(self):
        n = C.c_uint64(); r = C.c_int()
        _check(self.ctx.lib.plasship_cands_count(self.h, C.byref(n), C.byref(r)), "plasship_cands_count")
        return n.value

    def write(self, path):
        _check(self.ctx.lib.plasship_cands_write(self.ctx.h, self.h, self.qdb.h, os.fsencode(path)), "plasship_cands_write")

    def download(self):
        import numpy as np
        n = self.count()
        q = np.zeros(n, dtype=np.uint32); t = np.zeros(n, dtype=np.uint32); s = np.zeros(n, dtype=np.int32); d = np.zeros(n, dtype=np.uint16)
        _check(self.ctx.lib.plasship_cands_download(self.ctx.h, self.h, self.qdb.h, self.tdb.h, q.ctypes.data, t.ctypes.data, s.ctypes.data, d.ctypes.data), "plasship_cands_download")
        return q, t, s, d

    def free(self):
        if self.h:
            self.ctx.lib.plasship_cands_free(self.ctx.h, self.h); self.h = P()


class Alignments:
    def __init__(self, ctx, h, qdb, tdb):
        self.ctx, self.h, self.qdb, self.tdb = ctx, h, qdb, tdb   # keeps the DBs alive (ids -> keys)

    def count(self):
        n = C.c_uint64()
        _check(self.ctx.lib.plasship_alns_count(self.h, C.byref(n)), "plasship_alns_count")
        return n.value

    def write(self, path):
        _check(self.ctx.lib.plasship_alns_write(self.ctx.h, self.h, os.fsencode(path)), "plasship_alns_write")

    def download(self):
        n = self.count()
        arr = (AlnRecord * max(n, 1))()
        _check(self.ctx.lib.plasship_alns_download(self.ctx.h, self.h, arr), "plasship_alns_download")
        return arr[:n]

    def free(self):
        if self.h:
            self.ctx.lib.plasship_alns_free(self.ctx.h, self.h); self.h = P()
