set -o pipefail
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-base st1 st2 st1r3 st2r3 pr1 base}; do
  echo "== $v"
  if [ $v = base ]; then timeout -k 10 200 python tools/time_predict.py; else HSR_LIBRARY=$PWD/tools/dbg/libhsr_$v.so timeout -k 10 200 python tools/time_predict.py; fi
done
