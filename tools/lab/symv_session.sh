timeout 600 python -m pytest tests/test_eigh.py tests/test_device_kernels.py -m gpu -x -q 2>&1 | tail -2
for w in 0 256 512 768 1024 2048; do echo "== 12288 wgs=$w"; EIGH_SYMV_WGS=$w timeout 300 python tools/eigh_only.py 12288 3 2>&1 | tail -1; done
for w in 0 256 512 1024; do echo "== 8192 wgs=$w"; EIGH_SYMV_WGS=$w timeout 300 python tools/eigh_only.py 8192 3 2>&1 | tail -1; done
timeout 300 python tools/panel_bench.py 12288 16 2>&1 | head -1 | cut -c1-200
