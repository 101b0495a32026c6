cd $GRAFT_REPO_ROOT
for mode in pull auto; do
  if [ $mode = pull ]; then export QATZIP_AMD_K1=pull; else unset QATZIP_AMD_K1; fi
  echo "== parse kernel: $mode"
  timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 1
  timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 16
  timeout 60 ./build/var/bt_sweep perfmt 2 65536 2 64
  timeout 60 ./build/var/bt_sweep perfmt 16 524288 2 1
done
