#!/bin/bash
for v in "" p2 p3 p4 p6 m4 m10 p3m4 p3m10; do L=""; [ -n "$v" ] && L=$PWD/deplex_amd/lib/variants/$v.so; echo "== variant [$v]"; DPX_LIB_PATH=$L timeout -k 10 300 python tools/region_time.py 4k 1280 2>&1 | grep -v amdgpu.ids; done
