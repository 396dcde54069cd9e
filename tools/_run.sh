bash tools/final_profile.sh r02_mid 2>&1 | tail -15
