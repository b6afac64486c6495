#!/usr/bin/env python3
"""Per-kernel mean of the PMC counters in a rocprofv3 --pmc result database (rocpd sqlite).
Usage: tools/pmc_dump.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def main(path, needle=''):
    db = sqlite3.connect(path)
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in names if 'pmc' in t.lower() or 'counter' in t.lower()]
    print('tables/views with counters:', pmc)
    view = 'counters_collection' if 'counters_collection' in names else (pmc[0] if pmc else None)
    if view is None:
        print('no counter table; all tables:', names)
        return
    cols = [r[1] for r in cur.execute(f'pragma table_info({view})')]
    print(view, 'columns:', cols)
    kcol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else None)
    ccol = 'counter_name' if 'counter_name' in cols else None
    vcol = 'value' if 'value' in cols else ('counter_value' if 'counter_value' in cols else None)
    if not (kcol and ccol and vcol):
        for row in cur.execute(f'select * from {view} limit 5'):
            print(row)
        return
    dcol = 'dispatch_id' if 'dispatch_id' in cols else None
    # a counter is reported per dimension instance (XCD / channel): sum per dispatch first
    if dcol:
        q = (f"select {kcol}, {ccol}, count(*), avg(v), min(v), max(v) from (select {kcol}, {ccol}, {dcol}, sum({vcol}) as v "
             f"from {view} where {kcol} like ? group by {kcol}, {ccol}, {dcol}) group by {kcol}, {ccol} order by avg(v) desc")
    else:
        q = (f"select {kcol}, {ccol}, count(*), avg({vcol}), min({vcol}), max({vcol}) from {view} where {kcol} like ? "
             f"group by {kcol}, {ccol} order by avg({vcol}) desc")
    for k, cn, cnt, avg, mn, mx in cur.execute(q, ('%' + needle + '%',)):
        print(f'{k[:70]:70s} {cn:14s} dispatches={cnt:6d} mean={avg:.6g} min={mn:.6g} max={mx:.6g}')


if __name__ == '__main__':
    main(*sys.argv[1:3])
