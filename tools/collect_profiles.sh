#!/bin/bash
# Copy the summaries of an evidence session (gpurun_out/<tag>, written by tools/gpu_session.sh) into profiles/ under the
# round's names.  usage: bash tools/collect_profiles.sh <tag> <round prefix, e.g. r03>
TAG=$1; RP=${2:-r03}; S=gpurun_out/$TAG; P=profiles
cp $S/session.log $P/${RP}_session.log
cp $S/bench_line.json $P/${RP}_bench_line.json
for k in bench eigh davidson_loop block_iter optimizer_step; do cp $S/${k}_kernel_stats.md $P/${RP}_${k}_kernel_stats.md; done
cp $S/opt_step_timeline.txt $P/${RP}_opt_step_timeline.txt; cp $S/emt_step_timeline.txt $P/${RP}_emt_step_timeline.txt
cp $S/dav_iter_timeline.txt $P/${RP}_dav_iter_timeline.txt
cp $S/block_iter_timeline.txt $P/${RP}_block_iter_timeline.txt
{ echo "# configs[3] in lockstep cohorts (session $TAG): tools/emt_ensemble.py — t<T>: T host threads, one member each at a time; c<W>x<T>: T issuing threads, each advancing a cohort of W members (csrc/cohort.hip)"; echo '```'; cat $S/emt_cohorts.log | cut -c1-400; echo '```'; echo; echo "## GPU busy fraction of one cohort of 8 (rocprofv3 --kernel-trace of tools/emt_ensemble.py 8 c8; tools/cohort_busy.py over the launches of the timed pass; the profiler slows the host side)"; echo '```'; cat $S/cohort_busy.txt | cut -c1-200; echo '```'; echo; echo "## launches asked for by the members / issued after merging, by kernel body; microseconds of member host code in front of each park (SELLA_COHORT_TRACE=2; warm-up pass included)"; echo '```'; cat $S/cohort_by_kernel.log | cut -c1-160; echo '```'; } > $P/${RP}_cohorts.md
cp $S/member1_kernel_stats.md $P/${RP}_member_search_kernel_stats.md; cp $S/cohort8_kernel_stats.md $P/${RP}_cohort8_kernel_stats.md
{ echo "# tridiagonalisation at 3N = 3072 by trailing size (session $TAG): blocked chain above eigh_upd_max = 1024 rows, one launch per column below"; echo; cat $S/eigh_by_m.txt; echo; echo '## eigh wall time by switch-over size'; echo '```'; cat $S/eigh_switch.log; echo '```'; } > $P/${RP}_eigh_by_m.md
{ echo "# PMC passes (rocprofv3 --pmc <counter> with kernel dispatch tracing only, one counter group per run; tools/gpu_session.sh $TAG at the evidence head)"; echo;
  echo "Units: FETCH_SIZE / WRITE_SIZE in KiB as reported (raw). On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section): multiply FETCH by 2. Infinity-Cache hits are included."; echo;
  echo '## eigensolver, `tools/eigh_only.py 3072 1` — matvec kernel'; echo '```'; cat $S/pmc_eigh_FETCH_SIZE.txt $S/pmc_eigh_WRITE_SIZE.txt; echo '```'; echo;
  echo '## trailing rank-2k update of the tridiagonalisation (`rank2k_stream_fixed_kernel<16>`: every load issued before the first MFMA)'; echo '```'; cat $S/pmc_rank2k_FETCH.txt $S/pmc_rank2k_WRITE.txt; echo '```';
  echo "Same traffic as the generic kernel it replaced (session r03A: 2 x 14045 + 24384 KiB per mean dispatch in 24.4 us = 2.2 TB/s, 157 MB in 56.7 us = 2.8 TB/s for the largest; now 17 us mean = 3.2 TB/s, 36.7 us = 4.25 TB/s for the largest, durations in ${RP}_eigh_kernel_stats.md): the bytes were never the problem, the dependent operand loads were.  A plain in-place stream over the same block: 6.5 TB/s (r03_rmw_lab.log)."; echo;
  echo '## back-transformation (wy_apply_mfma_kernel), L2 requests, hits / misses and fetched bytes'; echo '```'; cat $S/pmc_wy_l2req.txt $S/pmc_wy_l2hit.txt $S/pmc_wy_fetch.txt; echo '```';
  echo "64 reflectors per block (n >= 2560; wy_apply_mfma64_kernel): 1.49e8 requests x 128 B = 19.1 GB through L2 in 2.50 ms = 7.6 TB/s (hit rate 86 %), FETCH x 2 = 3.7 GB past L2, against 0.23 GB algorithmic (read X once, write once, reflectors once); 2 n^3 = 5.8e10 flop in 2.50 ms = 23.2 TFLOP/s = 0.295 of the fp64 MFMA peak.  The 32-reflector kernel it replaced at this size (sessions up to r03D): 1.77e8 requests = 22.7 GB in 3.16 ms = 7.2 TB/s, hit rate 82 %, 6.6 GB past L2, 18.3 TFLOP/s = 0.23 — L2-bandwidth bound either way (4 / 8 / 16 wavefronts per workgroup measured equal); the 64-blocks halve the passes over X, the reflector stream stays."; echo;
  echo '## matrix-core utilisation (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)'; echo '```'; cat $S/pmc_mfma.txt; echo '```'; } > $P/${RP}_pmc.md
{ echo "# block Davidson iteration (tools/block_iter.py, 3N = 12288, block 16), pipelined driver (default) against the general loop (option bd_pipeline)"; echo '```'; cat $S/block_iter.log; echo '```'; } > $P/${RP}_block_iter.md
{ echo "# threads, processes and RCCL on one GPU (session $TAG)"; echo; echo '## ensemble leg on host threads of ONE process, searches inside the library (tools/ensemble_threads.py: EnsembleThreads, one persistent device context per thread)'; echo '```'; cat $S/threads.log; echo '```'; echo '## the same with the general driver (SELLA_LIBRARY_SEARCH=0: ~10,000 host-language calls per member, interpreter lock held in between)'; echo '```'; cat $S/threads_general.log; echo '```'; echo '## one member, where its time goes (tools/ens_profile.py)'; echo '```'; grep "seconds per member\|update_H n=768\|structured eigen" $S/ens_profile.log | tail -8; sed -n 1,18p $S/ens_profile.log; echo '```'; echo '## ensemble leg: worker processes (tools/ensemble_probe.py; workers run with HSA_ENABLE_SDMA=0)'; echo '```'; cat $S/probe.log; echo '```'; echo '## fine-grained library calls from N Python threads, one context each (tools/thread_scaling.py)'; echo '```'; cat $S/thread_scaling.log; echo '```'; echo '## RCCL: one rank (tools/rccl_smoke.py)'; echo '```'; cat $S/rccl.log; echo '```'; echo '## RCCL: two ranks on the one GPU of the box (tools/rccl_two_ranks_one_gpu.py)'; echo '```'; cat $S/rccl2.log; echo '```'; } > $P/${RP}_threads_procs_rccl.md
{ echo "# eigensolver beyond the Infinity Cache (session $TAG): symmetric-aware trailing matvec + triangle-only trailing update from 5120 trailing rows on, 64-reflector blocks in the back-transformation from n = 4096 on"; echo; echo '## wall time per eigh (tools/eigh_only.py)'; echo '```'; cat $S/eigh_large.log; echo '```'; echo; echo '## kernels of one eigh at 3N = 12288'; echo; sed -n 3,22p $S/eigh12288_kernel_stats.md; echo; echo '## per-column kernels by trailing size (tools/trd_by_m.py)'; echo; cat $S/eigh12288_by_m.txt; } > $P/${RP}_eigh_large.md
{ echo "# optimizer step and configs[1] timings (session $TAG)"; for f in opt_3072 emt geodesic dav_time; do echo; echo "## $f.log"; echo '```'; cat $S/$f.log; echo '```'; done; } > $P/${RP}_timings.md
python3 - "$S" "$RP" <<'PY'
import json, re, sys
S = sys.argv[1]
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
txt = open(S + '/pmc_eigh_FETCH_SIZE.txt').read() + open(S + '/pmc_eigh_WRITE_SIZE.txt').read()
# the matvec is a family of template instances (trd_gemv_kernel<NCH>): dispatch-weighted mean over all of them
rows = re.findall(r'(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+) mean=([0-9.e+]+)', txt)
vals = {k: sum(int(d) * float(v) for kk, d, v in rows if kk == k) / sum(int(d) for kk, d, v in rows if kk == k) for k in ('FETCH_SIZE', 'WRITE_SIZE')}
t = d['trd_gemv_kernel']
t['FETCH_SIZE_kb_mean_raw'], t['WRITE_SIZE_kb_mean_raw'] = vals['FETCH_SIZE'], vals['WRITE_SIZE']
t['bytes_per_launch'] = (2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024
t['source'] = re.sub(r'session r0\d\w?', 'session ' + S.split('/')[-1], re.sub(r'profiles/r0\d_pmc\.md', 'profiles/%s_pmc.md' % sys.argv[2], t['source']))
nd = sum(int(d) for kk, d, v in rows if kk == 'FETCH_SIZE')
t['dispatches'] = nd          # blocked chain only: trailing blocks of n-1 .. n-nd rows
t['algorithmic_bytes_per_launch'] = round(sum(8 * m * m + 16 * m for m in range(t['n'] - nd, t['n'])) / nd)
sess = S.split('/')[-1]
def grab(path):
    out = {}
    for line in open(path):
        m = re.search(r'(\S+)\s+dispatches=\s*(\d+) mean=([0-9.e+]+)', line)
        if m:
            nm = re.search(r'(\w+_kernel(?:<[^>]*>)?)', line)
            out[m.group(1)] = (int(m.group(2)), float(m.group(3)), nm.group(1) if nm else '')
    return out
try:
    r = {**grab(S + '/pmc_rank2k_FETCH.txt'), **grab(S + '/pmc_rank2k_WRITE.txt')}
    k = d['rank2k_stream_kernel']
    k['dispatches'] = r['FETCH_SIZE'][0]
    k['FETCH_SIZE_kb_mean_raw'], k['WRITE_SIZE_kb_mean_raw'] = r['FETCH_SIZE'][1], r['WRITE_SIZE'][1]
    k['bytes_per_launch'] = (2 * r['FETCH_SIZE'][1] + r['WRITE_SIZE'][1]) * 1024
    k.pop('mean_duration_us', None)
    k['source'] = 'profiles/%s_pmc.md (session %s); durations in profiles/%s_eigh_kernel_stats.md' % (sys.argv[2], sess, sys.argv[2])
    w = {**grab(S + '/pmc_wy_l2req.txt'), **grab(S + '/pmc_wy_l2hit.txt'), **grab(S + '/pmc_wy_fetch.txt')}
    key = 'wy_apply_mfma_kernel'
    d[key] = {'n': 3072, 'dispatches': w['FETCH_SIZE'][0], 'kernel': w['FETCH_SIZE'][2], 'FETCH_SIZE_kb_raw': w['FETCH_SIZE'][1],
              'bytes_fetched': 2 * w['FETCH_SIZE'][1] * 1024, 'TCP_TCC_READ_REQ': w['TCP_TCC_READ_REQ_sum'][1],
              'TCC_HIT': w['TCC_HIT_sum'][1], 'TCC_MISS': w['TCC_MISS_sum'][1], 'algorithmic_bytes': 3 * 8 * 3072 * 3072,
              'source': 'profiles/%s_pmc.md (session %s); duration in profiles/%s_eigh_kernel_stats.md' % (sys.argv[2], sess, sys.argv[2])}
except Exception as e:                                   # noqa: BLE001
    print('pmc_traffic.json: rank2k / wy entries not refreshed:', e)
json.dump(d, open(p, 'w'), indent=1)
print(vals)
PY
