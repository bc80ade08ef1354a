cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in spvcnn18:fuse:f32 spvcnn18:fuse:bf16 cylinder:fuse:f32 cylinder:fuse:bf16 rpvnet34:fuse:f32 rpvnet34:fuse:bf16 spvcnn18:reference:f32 rpvnet34:reference:f32; do
  tag=g_$(echo $s | tr ':+' '__')
  timeout 600 bash tools/profile_model.sh $tag $s > gpurun_out/${tag}.log 2>&1; grep "total" gpurun_out/${tag}_step_budget.md
done
