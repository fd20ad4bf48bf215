bash tools/run_checks_r05.sh
