#!/usr/bin/env python3
"""Canonicalise an MMseqs-format DB: one data file, entries in key order, key-sorted index.
Usage: dbcanon.py IN OUT"""
import sys, os

def canon(src, dst):
    if os.path.exists(src):
        data = open(src, 'rb').read()
    else:
        data = b''; i = 0
        while os.path.exists('%s.%d' % (src, i)):
            data += open('%s.%d' % (src, i), 'rb').read(); i += 1
    ent = []
    for n, line in enumerate(open(src + '.index', 'rb')):
        k, o, l = line.split()[:3]
        ent.append((int(k), n, int(o), int(l)))
    ent.sort()
    off = 0
    with open(dst, 'wb') as fd, open(dst + '.index', 'wb') as fi:
        for k, _, o, l in ent:
            fd.write(data[o:o + l]); fi.write(b'%d\t%d\t%d\n' % (k, off, l)); off += l
    open(dst + '.dbtype', 'wb').write(open(src + '.dbtype', 'rb').read()[:4])

if __name__ == '__main__':
    canon(sys.argv[1], sys.argv[2])
