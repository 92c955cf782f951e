#!/bin/bash
# Timeline of one iteration of the TRAINING DRIVER (loader + GPU augmentation + step):  scripts/timeline_driver.sh <tag> [step index]
TAG=${1:-tld}; WHICH=${2:-18}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 $ROOT/train_chaos.py --session=Experiment --tag=tl --root /tmp/tl_root --synthetic 512 --epoch 2 --max_iters 12 --batch_size 32 --image_size 256 --num_workers 4 --do_loss_ent --do_decoder_consistency --do_aux_path --do_memory --gpu_augment > "$OUT/kt.log" 2>&1
python3 $ROOT/scripts/timeline.py "$OUT/kt" "$OUT/timeline.tsv" $WHICH
rm -rf "$OUT/kt" /tmp/tl_root
