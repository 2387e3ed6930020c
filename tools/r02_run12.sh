O=gpurun_out/r02n; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
python3 - <<'P'
import sys; sys.path.insert(0,'.')
from firework_amd import scenes, _lib
for n in ("C3_suzanne","C5_part2_all","C1_random_spheres"):
    s,r=scenes.config(n,64,36,1); st=r.render_full(s).stats
    print(n,"tlas_depth",st["reserved"]>>16,"blas_depth",st["reserved"]&0xffff)
P
