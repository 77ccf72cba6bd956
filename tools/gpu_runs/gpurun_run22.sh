export LD_LIBRARY_PATH=$PWD/bitmagic_amd/lib:/opt/rocm/lib
./oracle/_ref/test_adapter_ref; echo rc=$?
