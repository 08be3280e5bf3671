#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in km,kn mk,kn; do
  echo "layout $L random w4"; WAVES=4 bash tools/h16_abl.sh "0 1 2 3 4 5" $L
  echo "layout $L zeros w4"; WAVES=4 bash tools/h16_abl.sh "0 1 4 5" $L --zeros
  echo "layout $L random w8"; WAVES=8 bash tools/h16_abl.sh "0" $L
  echo "layout $L zeros w8"; WAVES=8 bash tools/h16_abl.sh "0" $L --zeros
done
