bash tools/final_profile.sh r02_final2 2>&1 | tail -14
