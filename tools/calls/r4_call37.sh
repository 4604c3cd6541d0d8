# round 4, call 37: band height of the XCD tile walk for the 256 x 128 NT plane tile (A: 8 row tiles, G4: 4, G2: 2) -- per-site times in the step
cd /root/repo
for v in A G4 G2 A G4 G2; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo "== $v"; python tools/step_plane_profile.py 2>/dev/null | grep "plane products\|tile256128"; done
cp abso/libvbg_A.so vibertgrid-pytorch_amd/libvbg.so
