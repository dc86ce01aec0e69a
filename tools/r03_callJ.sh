set -e
OUT=gpurun_out/r03j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -c "
import cProfile,pstats,sys,runpy,io
sys.argv=['tools/fit_q1422.py']
pr=cProfile.Profile()
pr.enable()
try:
    runpy.run_path('tools/fit_q1422.py', run_name='__main__')
finally:
    pr.disable()
    s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); open('$OUT/fit_profile_cum.txt','w').write(s.getvalue())
    s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(35); open('$OUT/fit_profile_tot.txt','w').write(s.getvalue())
" > $OUT/fit.log 2>&1
tail -1 $OUT/fit.log | cut -c1-300
