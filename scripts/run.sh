#!/usr/bin/env bash
# Same job as the reference's scripts/run.sh (default scene, five mode flags ->
# output/plain/output<suffix>.ppm), on top of bin/ESCViewer2021.  Every mode renders on the GPU;
# the flags are kept so existing command lines keep working.
#   MODELS_DIR  directory holding cornell/CornellBox-Original.obj (default: the reference tree,
#               else the committed fixture scene tests/golden/scenes/one.obj is used)
current_dir=$( cd "$( dirname "${BASH_SOURCE[0]}" )" && pwd )
ROOT=$( cd "$current_dir/.." && pwd )
OUT_DIR=${PROJECT_OUT_DIR:-$ROOT/output}/plain
MODELS_DIR=${MODELS_DIR:-/root/reference/src/models}
program=$ROOT/bin/ESCViewer2021
mkdir -p "$OUT_DIR"

model="$MODELS_DIR/cornell/CornellBox-Original.obj"
eye="0,1,2"
look="0,1,0"
if [ ! -f "$model" ]; then
  model="$ROOT/tests/golden/scenes/one.obj"; eye="0,1,3"
  echo "bundled models not found, using $model"
fi

run() {
  outfile="$OUT_DIR/output${suffix}.ppm"
  rm -f "$outfile"
  echo "$program -m '$model' -v '$eye' -l '$look' ${options} -o $outfile"
  "$program" -m "$model" -v "$eye" -l "$look" -o "$outfile" ${options} || exit 1
  [ -f "$outfile" ] && echo "Created new rendering at $outfile" || echo "WARNING: no file at $outfile"
}

echo "running with no options"; suffix="sequential"; options=""; run
for options in "--thread" "--bvh" "--bvh --thread" "--ispc"; do
  suffix=$(echo $options | sed 's/[ -]//g')
  echo "running with $options"
  run
done
md5sum "$OUT_DIR"/output*.ppm
