mkdir -p gpurun_out
./tests/cpp/_bin/test_facade > gpurun_out/cpp_tests.log 2>&1; echo "facade rc=$?" >> gpurun_out/cpp_tests.log
./oracle/_ref/test_adapter_ref >> gpurun_out/cpp_tests.log 2>&1; echo "adapter rc=$?" >> gpurun_out/cpp_tests.log
tail -4 gpurun_out/cpp_tests.log
timeout 600 ./oracle/_ref/bench_adapter_upload > gpurun_out/bench_adapter_upload.log 2>&1; tail -4 gpurun_out/bench_adapter_upload.log
