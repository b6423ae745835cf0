B=tools/bin/mmv3_lab_nw
run() { echo "## $*"; env "$@" timeout 60 $B 0 3 2>&1 | grep "launch form" | awk '{print $6}' | tr '\n' ' '; echo; }
run X=1
run MV2_NW_PAIR=16 MV2_NW_GRP=16 MV2_NW_SMALL4=16 MV2_NW_SMALL12=16
for v in 9 10 12 13 16; do run MV2_NW_SMALL4=$v; done
for v in 9 10 12 13 16; do run MV2_NW_SMALL12=$v; done
for v in 9 10 12 13 16; do run MV2_NW_GRP=$v; done
for v in 9 10 12 13 16; do run MV2_NW_PAIR=$v; done
