mkdir -p gpurun_out/r05o
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05o/smoke_direct.log 2>&1; tail -3 gpurun_out/r05o/smoke_direct.log
python __graft_entry__.py smoke > gpurun_out/r05o/smoke_main.log 2>&1; tail -12 gpurun_out/r05o/smoke_main.log
bash tools/dbg/r05n.sh
