cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_gpu.py -m gpu -q -k "geometries" -p no:cacheprovider 2>&1 | tail -3
for v in "--steps 6000" "--steps 3000 --lr-pose-end 5e-4" "--steps 3000 --w-corres 3e-2" "--steps 3000 --c2f 0.05 0.35" "--steps 6000 --lr-pose-end 3e-4 --w-corres 3e-2"; do
  echo "== $v"
  timeout 900 python tests/tools/registration_run.py --trainers hip --seeds 3 --eval-every 1000 --quiet $v --out gpurun_out/reg_tmp.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/reg_tmp.json'))
print('   final', [round(x,2) for x in d['final']['hip']['rot_err_deg_final']], 'psnr', [round(x,1) for x in d['final']['hip']['psnr_final']])
for r in d['runs']: print('   seed', r['seed'], [(c['step'], round(c['hip']['rot_err_deg'],2)) for c in r['curve']])"
done
