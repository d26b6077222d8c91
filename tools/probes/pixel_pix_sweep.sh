cd $GRAFT_REPO_ROOT
for pix in 256 512 1024; do echo "PIX=$pix"; SF_PIXEL_PIX=$pix python tools/pixel_probe.py 2>&1 | grep "form [12]"; done
