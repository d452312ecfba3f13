# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
.PHONY: build test test-gpu bench sanitize clean

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

clean:
	rm -f godot_atmosphere_shader_amd/libatmo_hip*.so oracle/*.so tools/valu_peak
