set -x
for leaf in 32 16 8 64; do echo "== 3072 leaf=$leaf"; EIGH_LEAF=$leaf timeout 300 python tools/eigh_only.py 3072 5 2>&1 | tail -2; done
for leaf in 32 16; do echo "== 768 leaf=$leaf"; EIGH_LEAF=$leaf timeout 300 python tools/eigh_only.py 768 5 2>&1 | tail -2; done
