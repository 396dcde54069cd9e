bash tools/final_profile.sh r02_final 2>&1 | tail -14
