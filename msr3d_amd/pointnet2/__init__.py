"""MI355X-native `pointnet2` package: `_ext` (the nine HIP ops), the autograd
wrappers (`pointnet2_utils`), the set-abstraction modules (`pointnet2_modules`)
and `pytorch_utils.SharedMLP`, mirroring
/root/reference/modules/third_party/pointnet2/."""
