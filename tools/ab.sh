#!/bin/bash
# run tools/one.py against each variant library:  tools/ab.sh "args for one.py" name...
args=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then lib=pathtracer_amd/libptcore.so; else lib=pathtracer_amd/variants/libptcore_$v.so; fi
  echo "== $v"
  PTCORE_LIB=$lib timeout -k 10 120 python3 -u tools/one.py $args 2>&1 | grep -v amdgpu.ids || exit 1
done
