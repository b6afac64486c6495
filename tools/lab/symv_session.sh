for nb in 16 32 24 8 48; do echo "== 3072 nb=$nb"; EIGH_NB=$nb timeout 300 python tools/eigh_only.py 3072 5 2>&1 | tail -2; done
for nb in 16 32; do echo "== 12288 nb=$nb"; EIGH_NB=$nb timeout 300 python tools/eigh_only.py 12288 2 2>&1 | tail -1; done
for nb in 16 32; do echo "== 768 nb=$nb"; EIGH_NB=$nb timeout 300 python tools/eigh_only.py 768 5 2>&1 | tail -1; done
