cd $GRAFT_REPO_ROOT
env | grep -i "HIP_\|HSA_\|GPU_\|AMD_\|ROC" | grep -v PATH
timeout 60 ./build/stream_overlap
