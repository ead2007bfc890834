#!/bin/bash
# Build container only: encodes small streams with the REFERENCE encoder (cc_encode.py --debug, i.e. the
# same short training samples/encode.py --debug runs) for the decoder presets the r03 fixtures did not
# cover (cfg/dec/intra/{mop,vhop}.cfg, cfg/dec/residue/{hop,mop,vlop}.cfg, cfg/dec/motion/mop.cfg), then
# dumps the reference DECODER's per-stage results with dump_reference.py.  Nothing here runs on the GPU box.
#
#   tests/golden/gen/encode_presets.sh <name> <input under /root/reference/test/data> <n_frames> <intra cfg> [<residue cfg> <motion cfg>]
set -e
NAME=$1; IN=$2; NF=$3; INTRA=$4; RES=${5:-lop}; MOT=${6:-lop}
# LMBDA (environment): the rate constraint, default the encoder's 1e-3; hq192 = LMBDA=0.00001 encode_presets.sh hq192 192x128_kodim15.png 1 lop
# (2.5 bpp: 12-15 % of the symbols take the wide 62-symbol windows, scale indices up to 2433 of 2560)
LMBDA_ARG=${LMBDA:+--lmbda=$LMBDA}
# <input>: a file name under /root/reference/test/data, or an absolute path (crops of the reference's test picture made by the caller)
case "$IN" in /*) INP=$IN;; *) INP=/root/reference/test/data/$IN;; esac
HERE=$(cd "$(dirname "$0")" && pwd)
export PYTHONDONTWRITEBYTECODE=1
export PYTHONPATH=$HERE/shims:/root/reference
W=/tmp/ccgen/$NAME
rm -rf $W; mkdir -p $W
cd /root/reference
# PPOS (environment): display indices of the P frames (cc_encode.py --p_pos), default: the last frame (hierarchical B in between);
# vid3_ldp = PPOS=1-2 encode_presets.sh vid3_ldp D-BQSquare-3frames_224x128_60p_yuv420_8b.yuv 3 lop lop lop   (I P P: a P frame predicted from a P frame)
if [ "$NF" -gt 1 ]; then PPOS="--p_pos=${PPOS:--1}"; else PPOS=""; fi
for ((k = 0; k < NF; k++)); do
    if [ $k -eq 0 ]; then R=/root/reference/cfg/dec/intra/$INTRA.cfg; else R=/root/reference/cfg/dec/residue/$RES.cfg; fi
    python3 cc_encode.py --input=$INP --workdir=$W/ --intra_pos=0 $PPOS --n_frames=$NF \
        --output=$W/$NAME.cool --coding_idx=$k --debug $LMBDA_ARG \
        --dec_cfg_residue=$R --dec_cfg_motion=/root/reference/cfg/dec/motion/$MOT.cfg > $W/enc$k.log 2>&1
done
ls -la $W/$NAME.cool
cp $W/$NAME.cool $HERE/../$NAME.cool
python3 $HERE/dump_reference.py $HERE/../$NAME.cool $NAME
