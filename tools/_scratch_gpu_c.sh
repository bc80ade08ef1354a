cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in spvcnn18:fuse:f32 spvcnn18:fuse:bf16 rpvnet34:fuse:f32 rpvnet34:fuse:bf16 cylinder:reference:f32 cylinder:reference:bf16; do
  tag=c_$(echo $s | tr ':+' '__')
  timeout 600 bash tools/profile_model.sh $tag $s > gpurun_out/${tag}.log 2>&1; tail -4 gpurun_out/${tag}.log | head -2
done
