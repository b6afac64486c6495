#!/usr/bin/env python3
"""Per-column kernel durations of the tridiagonalisation out of a rocprofv3 kernel trace (rocpd sqlite), binned by the
size m of the trailing block: the k-th matvec of a factorisation works on m = n - 1 - k rows.  Only the first eigh of
the trace is used.  Usage: tools/trd_by_m.py <results.db> [bin] [n]   (n: the matrix size; needed when the last columns
run inside trd_tail_lds_kernel and have no matvec launch of their own)"""
import sqlite3
import sys


def main(db_path, width=256, n_given=0):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute('select name, start, duration from kernels order by start').fetchall()
    kinds = {'trd_gemv_kernel': 'gemv', 'trd_symv_kernel': 'symv', 'trd_symv_finish_kernel': 'finish', 'trd_row_kernel': 'row'}
    seq, ncol = [], 0
    for name, _, dur in rows:
        kind = next((v for k, v in kinds.items() if k + '(' in name or k + '<' in name), None)
        if 'tridiag_tail_kernel' in name or 'trd_tail_lds_kernel' in name:
            break                                     # end of the first factorisation (the blocked chain may hand over earlier:
                                                      # trd_upd_kernel launches, tools/col_by_m.py, carry no matvec of their own)
        if kind is None:
            continue
        if kind in ('gemv', 'symv'):
            ncol += 1
        seq.append((kind, ncol, dur))
    n = n_given if n_given else ncol + 2
    bins = {}
    for kind, k, dur in seq:
        m = n - 1 - max(k, 1)
        b = bins.setdefault(m // width, {})
        t = b.setdefault(kind, [0, 0.0])
        t[0] += 1
        t[1] += dur
    print(f'n = {n}; mean microseconds per launch, by trailing size m (bins of {width})')
    print('| m | gemv | symv | finish | row | matvec GB/s (8 m^2 or 4 m^2 per launch) |')
    print('|---|---:|---:|---:|---:|---:|')
    for key in sorted(bins, reverse=True):
        b = bins[key]
        m_mid = key * width + width / 2
        cell = lambda kind: f'{b[kind][1] / b[kind][0] / 1e3:.2f}' if kind in b else ''
        rate = ''
        if 'symv' in b:
            rate = f'{4 * m_mid * m_mid / (b["symv"][1] / b["symv"][0]):.0f}'
        elif 'gemv' in b:
            rate = f'{8 * m_mid * m_mid / (b["gemv"][1] / b["gemv"][0]):.0f}'
        print(f'| {key * width}-{key * width + width - 1} | {cell("gemv")} | {cell("symv")} | {cell("finish")} | {cell("row")} | {rate} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
