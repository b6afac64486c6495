#!/usr/bin/env python3
"""Per-column durations of the one-launch-per-column tridiagonalisation (trd_upd_kernel) out of a rocprofv3 kernel trace
(rocpd sqlite), binned by the size m of the trailing block; LAST eigh of the trace (the first one pays first-touch).
Usage: tools/col_by_m.py <results.db> <n> [bin]"""
import sqlite3
import sys


def main(db_path, n, width=256):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute('select name, start, duration, grid_x, workgroup_x from kernels order by start').fetchall()
    runs, cur_run = [], []
    for name, _, dur, gx, wx in rows:
        if 'trd_upd_kernel' in name:
            cur_run.append((dur, gx, wx))
        elif 'tridiag_tail_kernel' in name or 'trd_tail_lds_kernel' in name:
            if cur_run:
                runs.append(cur_run)
            cur_run = []
    if cur_run:
        runs.append(cur_run)
    if not runs:
        print('no trd_upd_kernel launches in the trace')
        return
    seq = runs[-1]
    bins = {}
    first = len(seq) + 128                        # the chain ends where the LDS tail (last 128 columns) takes over
    for k, (dur, gx, wx) in enumerate(seq):
        m = first - k
        b = bins.setdefault(m // width, [0, 0.0, 0, 0])
        b[0] += 1
        b[1] += dur
        b[2] = max(b[2], gx // max(wx, 1))
        b[3] = wx
    tot = sum(d for d, _, _ in seq)
    print(f'n = {n}: {len(seq)} column launches, {tot / 1e6:.2f} ms in the kernel; mean microseconds per launch by trailing size m')
    print('| m | launches | us | workgroups | threads | 16 m^2 GB/s (block read and written) |')
    print('|---|---:|---:|---:|---:|---:|')
    for key in sorted(bins, reverse=True):
        cnt, t, g, wx = bins[key]
        mm = key * width + width / 2
        print(f'| {key * width}-{key * width + width - 1} | {cnt} | {t / cnt / 1e3:.2f} | {g} | {wx} | {16 * mm * mm / (t / cnt):.0f} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 256)
