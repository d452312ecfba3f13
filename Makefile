# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
.PHONY: build test test-gpu bench sanitize sanitize-host clean

build:
	python -c "import __graft_entry__ as g; g.build()"

test: build
	python -m pytest tests -x -q -m "not gpu"

test-gpu:           # on an MI355X
	python -m pytest tests -x -q -m gpu

bench:              # on an MI355X
	python bench.py

sanitize:           # ASan + UBSan run of the CPU oracle's tests
	$(MAKE) -C oracle sanitize
	LD_PRELOAD=$$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 ORACLE_SANITIZE=1 \
	    python -m pytest tests/test_oracle_kat.py tests/test_noise_cubemap.py -q -m "not gpu"

sanitize-host:      # ASan + UBSan over the HOST side of libatmo_hip.so (uniform table, per-frame constants, layout helpers, argument checks,
                    # the motion estimate) driven by the no-GPU tests; the device code is built as always (no GPU sanitizers on this pool)
	set -e; out=$$(python -m godot_atmosphere_shader_amd.build --sanitize | tail -1); lib=$${out%% *}; rt=$${out##* }; \
	LD_PRELOAD=$$rt ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 ATMO_HIP_LIB=$$lib \
	    python -m pytest tests/test_host_logic.py -q -m "not gpu" -k "not whole_quad and not bench"

clean:
	rm -f godot_atmosphere_shader_amd/libatmo_hip*.so oracle/*.so tools/valu_peak
