set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile_round.sh r04a softball > gpurun_out/r04a_pr_softball.log 2>&1
bash scripts/profile_round.sh r04a softcylinder > gpurun_out/r04a_pr_softcylinder.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/section_profile.py softball > gpurun_out/r04a_sections_softball.txt 2>&1
python3 scripts/section_profile.py softcylinder > gpurun_out/r04a_sections_softcylinder.txt 2>&1
python3 scripts/solver_stats.py softball > gpurun_out/r04a_solver_stats_softball.txt 2>&1
python3 scripts/solver_stats.py softcylinder > gpurun_out/r04a_solver_stats_softcylinder.txt 2>&1
python3 scripts/phase_time.py softball > gpurun_out/r04a_phase_time.txt 2>&1
python3 scripts/phase_time.py softcylinder >> gpurun_out/r04a_phase_time.txt 2>&1
python3 scripts/phase_time.py softbox >> gpurun_out/r04a_phase_time.txt 2>&1
echo done
