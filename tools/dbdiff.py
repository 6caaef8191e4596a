#!/usr/bin/env python3
"""Compare two MMseqs-format DBs on canonical form (key -> entry bytes), order independent.
Usage: dbdiff.py A B [--max 5]"""
import sys, os

def read_db(name):
    if os.path.exists(name):
        data = open(name, 'rb').read()
    else:
        data = b''; i = 0
        while os.path.exists('%s.%d' % (name, i)):
            data += open('%s.%d' % (name, i), 'rb').read(); i += 1
    ent = {}
    for line in open(name + '.index', 'rb'):
        k, o, l = line.split()[:3]
        ent.setdefault(int(k), []).append(data[int(o):int(o) + int(l)])
    dbtype = int.from_bytes(open(name + '.dbtype', 'rb').read()[:4], 'little')
    return dbtype, ent

def main():
    a, b = sys.argv[1], sys.argv[2]
    mx = 5
    ta, ea = read_db(a); tb, eb = read_db(b)
    bad = 0
    if ta != tb:
        print('dbtype differs: %d vs %d' % (ta, tb)); bad += 1
    ka, kb = set(ea), set(eb)
    if ka != kb:
        print('key sets differ: only A %d, only B %d (e.g. %s / %s)' % (len(ka - kb), len(kb - ka), sorted(ka - kb)[:5], sorted(kb - ka)[:5])); bad += 1
    nd = 0
    for k in sorted(ka & kb):
        if ea[k] != eb[k]:
            nd += 1
            if nd <= mx:
                print('key %d differs:\n  A=%r\n  B=%r' % (k, ea[k][0][:400], eb[k][0][:400]))
    print('%d keys, %d differing entries' % (len(ka & kb), nd))
    sys.exit(1 if (bad or nd) else 0)

if __name__ == '__main__':
    main()
