#!/bin/bash
# usage: run.sh  (on a GPU box)
cd "$(dirname "$0")"
for fl in 3 1 0; do
  rm -f /tmp/p2p_handle.bin
  timeout -k 5 30 ./probe A /tmp/p2p_handle.bin $fl &
  pa=$!
  timeout -k 5 30 ./probe B /tmp/p2p_handle.bin $fl
  wait $pa
  echo "flags=$fl exit A=$?"
done
