for g in 0 2048; do echo "gs_small=$g"; SELLA_GS_SMALL=$g SELLA_DEBUG_TIMING=1 python tools/emt_member_time.py 2>&1 | grep -E "davidson: k=|member" | tail -4; done
