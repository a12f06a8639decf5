cd $GRAFT_REPO_ROOT
S="32,384,384,7,1,3,5568,0,1 32,768,768,7,1,3,696,0,1 32,512,512,7,1,3,696,0,1 32,256,256,7,1,3,5568,0,1 32,128,128,7,1,3,22272,0,1"
for tm in 0 4 3 2; do echo "== NC_TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done
