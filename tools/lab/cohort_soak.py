"""Soak test of the cohort scheduler on the device: the same 8 EMT members through cohorts of several shapes, fibers and
member threads, many passes — every pass must reproduce the first bit for bit (a race between member threads and the
issuing thread would show up as a differing geometry or a hang).   usage: cohort_soak.py [passes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import EmtSlabMember  # noqa: E402
from sella_amd.ensemble import EnsembleCohort, EnsembleCohorts, run_ensemble  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fac = EmtSlabMember()
ref = None
t0 = time.time()
for shape in ((8, 1, False), (8, 1, True), (4, 2, True), (3, 3, True), (2, 4, True), (5, 2, False)):
    w, t, mt = shape
    with (EnsembleCohort(w, member_threads=mt) if t == 1 else EnsembleCohorts(w, t, member_threads=mt)) as co:
        for p in range(passes):
            res = run_ensemble(fac, 8, fmax=0.0, steps=12, sella_kwargs=EmtSlabMember.SELLA_KW, cohort=co)
            key = (res['summary'].tobytes(), b''.join(x.tobytes() for x in res['positions']))
            if ref is None:
                ref = key
            assert key == ref, ('pass differs', shape, p)
    print('shape width %d x %d issuing threads, member threads %s: %d passes identical' % (w, t, mt, passes), flush=True)
print('ok in %.1f s' % (time.time() - t0))
