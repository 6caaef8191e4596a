#!/usr/bin/env python3
"""W ranks of the NATIVE communicator (include/plasship_rccl.h, plass_amd/csrc/comm_rccl.hip) in one process on one GPU — test
infrastructure, started by tests/test_gpu_sharded.py in a child process with PLASSHIP_RCCL_LIB pointing at tests/tools/librccl_stub.so
(the nccl* symbols over an in-process rendezvous, tests/tools/rccl_stub.cpp; the library reads the variable once per process).

    native_comm_ranks.py <world> <golden dir> <out dir> <case>

Every rank is a thread with its own plasship context; the communicator is created with plasship_rccl_comm_create on the id rank 0
made (the collective a C++ host would run), the chains below are the ones of the LocalGroup tests.  Each rank writes its DBs as
<out>/r<rank>_<name>; the parent compares them with the golden DBs / the single-context result.  Prints NATIVE_OK <bytes> <calls>."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, golden, out, case = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
    import plass_amd
    from plass_amd import synth
    from plass_amd.shard import RcclComm, rccl_unique_id
    from test_gpu_parity import gd_km_params, gd_rs_params, km_params, nucl_as_params
    uid = rccl_unique_id()
    ctxs = [plass_amd.Context(0) for _ in range(world)]
    stats, errs = [None] * world, [None] * world
    synth_db = synth.protein_fragment_db(40000, seed=5) if case == "synthetic" else None

    def chain(rank, ctx):
        P = lambda n: os.path.join(out, "r%d_%s" % (rank, n))
        if case == "aa":
            db = ctx.read_seqdb(os.path.join(golden, "aa", "seq_0"))
            for it in range(3):
                c, _ = ctx.kmermatcher(db, km_params(it))
                a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.9))
                c.write(P("pref_%d" % it)); a.write(P("aln_%d" % it))
                db, _ = ctx.assembleresults(db, a, plass_amd.AssembleParams(min_seq_id=0.9))
                db.write(P("seq_%d" % (it + 1)))
        elif case == "synthetic":
            db = ctx.upload_seqdb(*synth_db, 0)
            for it in range(3):
                c, _ = ctx.kmermatcher(db, km_params(it))
                a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.9))
                db, _ = ctx.assembleresults(db, a, plass_amd.AssembleParams(min_seq_id=0.9))
                db.write(P("seq_%d" % (it + 1)))
        elif case == "nucl":
            db = ctx.read_seqdb(os.path.join(golden, "nucl", "seq_0"))
            for it in range(2):
                c, _ = ctx.kmermatcher(db, km_params(it, nucl=True))
                a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.99))
                db, _ = ctx.assembleresults(db, a, nucl_as_params())
                db.write(P("seq_%d" % (it + 1)))
        elif case == "guided":
            s = os.path.join(golden, "guided")
            aa = ctx.read_seqdb(s + "/aa_0"); nu = ctx.read_seqdb(s + "/nucl_0")
            for it in range(2):
                c, _ = ctx.kmermatcher(aa, gd_km_params())
                a, _ = ctx.rescorediagonal(aa, aa, c, gd_rs_params())
                na, _ = ctx.proteinaln2nucl(nu, aa, a)
                nu, aa, _ = ctx.guidedassembleresults(nu, aa, na)
                nu.write(P("nucl_%d" % (it + 1))); aa.write(P("aa_%d" % (it + 1)))
        else:
            raise SystemExit("unknown case " + case)

    def work(rank):
        try:
            comm = RcclComm(ctxs[rank], rank, world, uid)          # collective: returns when every rank has joined
            try:
                chain(rank, ctxs[rank])
                stats[rank] = comm.stats()
            finally:
                comm.destroy()
        except BaseException as e:
            errs[rank] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r, e in enumerate(errs):
        if e is not None:
            print("rank %d failed: %r" % (r, e))
    if any(e is not None for e in errs):
        raise SystemExit(1)
    for c in ctxs:
        c.close()
    print("NATIVE_OK", sum(s[0] for s in stats), sum(s[2] for s in stats))


if __name__ == "__main__":
    main()
