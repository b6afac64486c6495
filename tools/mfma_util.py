#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` database:
busy cycles summed over the device, divided by (SIMDs x kernel duration x engine clock).
Usage: tools/mfma_util.py <results.db> <kernel-substring> [flops-per-dispatch]"""
import sqlite3
import sys

SIMDS = 256 * 4
CLOCK_GHZ = 2.4            # MI355X peak engine clock (MI355X_MICROARCH.md constants table)
PEAK_F64_TF = 78.6


def main(path, needle, flops=None):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select dispatch_id, counter_name, sum(value), count(*), max(end - start) from counters_collection "
        "where kernel_name like ? group by dispatch_id, counter_name", ('%' + needle + '%',)))
    per = {}
    for did, name, total, ninst, dur in rows:
        per.setdefault(did, {})[name] = (total, ninst, dur)
    busy, gui, durs = [], [], []
    for d in per.values():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
            busy.append(d['SQ_VALU_MFMA_BUSY_CYCLES'][0])
            durs.append(d['SQ_VALU_MFMA_BUSY_CYCLES'][2])
        if 'GRBM_GUI_ACTIVE' in d:
            # reported as ONE value per dispatch that already sums the 8 XCDs (19.3 'GHz' over a kernel = 8 x 2.4)
            tot, ninst, _ = d['GRBM_GUI_ACTIVE']
            gui.append(tot / (8.0 if ninst == 1 else ninst))
    if not busy:
        print('no dispatches of', needle)
        return
    n = len(busy)
    mb, md = sum(busy) / n, sum(durs) / n
    line = f'{needle}: dispatches={n} mean duration {md / 1e3:.1f} us, SQ_VALU_MFMA_BUSY_CYCLES mean {mb:.4g}'
    if gui:
        mg = sum(gui) / len(gui)
        line += f', GRBM_GUI_ACTIVE per XCD {mg:.4g} (= {mg / md:.2f} GHz over the kernel)'
        line += f', MFMA busy / (1024 SIMDs x GUI_ACTIVE) = {mb / (SIMDS * mg):.3f}'
    line += f', MFMA busy / (1024 SIMDs x duration x {CLOCK_GHZ} GHz) = {mb / (SIMDS * md * CLOCK_GHZ):.3f}'
    if flops:
        tf = float(flops) / (md * 1e-9) / 1e12
        line += f', {tf:.1f} TFLOP/s = {tf / PEAK_F64_TF:.3f} of the fp64 MFMA peak'
    print(line)


if __name__ == '__main__':
    main(*sys.argv[1:4])
