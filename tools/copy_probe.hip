// copy_probe: what a host-pointer entry pays for its transfers on this box.  Times hipMemcpyAsync of the benchmark's
// genotype matrix (6 MB up) and outputs (4.2 MB down) from / to pageable and pinned host memory, on the null stream and on a
// non-blocking stream, and a memcpy into a pinned staging buffer.  Build: hipcc --offload-arch=gfx950 -O2 -o copy_probe copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
	const size_t up = 10000ull * 150 * 4, down = 10000ull * (50 * 8 + 24);
	void *d_up, *d_down, *pin_up, *pin_down;
	CK(hipMalloc(&d_up, up)); CK(hipMalloc(&d_down, down));
	CK(hipHostMalloc(&pin_up, up, hipHostMallocDefault)); CK(hipHostMalloc(&pin_down, down, hipHostMallocDefault));
	std::vector<char> pg_up(up, 1), pg_down(down, 0);
	hipStream_t nb; CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
	struct Case { const char *name; void *host; void *dev; size_t n; hipMemcpyKind kind; hipStream_t st; };
	const Case cases[] = {
		{"H2D 6 MB pageable, null stream", pg_up.data(), d_up, up, hipMemcpyHostToDevice, 0},
		{"H2D 6 MB pageable, non-blocking stream", pg_up.data(), d_up, up, hipMemcpyHostToDevice, nb},
		{"H2D 6 MB pinned, non-blocking stream", pin_up, d_up, up, hipMemcpyHostToDevice, nb},
		{"D2H 4.2 MB pageable, null stream", pg_down.data(), d_down, down, hipMemcpyDeviceToHost, 0},
		{"D2H 4.2 MB pageable, non-blocking stream", pg_down.data(), d_down, down, hipMemcpyDeviceToHost, nb},
		{"D2H 4.2 MB pinned, non-blocking stream", pin_down, d_down, down, hipMemcpyDeviceToHost, nb},
	};
	for (const Case &c : cases) {
		double best = 1e9, call = 0;
		for (int rep = 0; rep < 12; rep++) {
			CK(hipDeviceSynchronize());
			const double t0 = now();
			if (c.kind == hipMemcpyHostToDevice) CK(hipMemcpyAsync(c.dev, c.host, c.n, c.kind, c.st));
			else CK(hipMemcpyAsync(c.host, c.dev, c.n, c.kind, c.st));
			const double t1 = now();
			CK(hipStreamSynchronize(c.st));
			const double t2 = now();
			if (rep >= 2 && t2 - t0 < best) { best = t2 - t0; call = t1 - t0; }
		}
		printf("%-44s %7.3f ms (the call itself returns after %7.3f ms)  %6.1f GB/s\n", c.name, best * 1e3, call * 1e3, c.n / best / 1e9);
	}
	{
		double best = 1e9;
		for (int rep = 0; rep < 12; rep++) { const double t0 = now(); memcpy(pin_up, pg_up.data(), up); const double t = now() - t0; if (rep >= 2 && t < best) best = t; }
		printf("%-44s %7.3f ms  %6.1f GB/s\n", "memcpy 6 MB pageable -> pinned (one thread)", best * 1e3, up / best / 1e9);
		best = 1e9;
		for (int rep = 0; rep < 12; rep++) { const double t0 = now(); memcpy(pg_down.data(), pin_down, down); const double t = now() - t0; if (rep >= 2 && t < best) best = t; }
		printf("%-44s %7.3f ms  %6.1f GB/s\n", "memcpy 4.2 MB pinned -> pageable", best * 1e3, down / best / 1e9);
	}
	{	// the five output arrays as five copies against one
		double best = 1e9;
		const size_t part[5] = {40000, 40000, 80000, 80000, 4000000};
		for (int rep = 0; rep < 12; rep++) {
			CK(hipDeviceSynchronize());
			const double t0 = now();
			size_t off = 0;
			for (int i = 0; i < 5; i++) { CK(hipMemcpyAsync(pg_down.data() + off, (char *)d_down + off, part[i], hipMemcpyDeviceToHost, nb)); off += part[i]; }
			CK(hipStreamSynchronize(nb));
			const double t = now() - t0;
			if (rep >= 2 && t < best) best = t;
		}
		printf("%-44s %7.3f ms\n", "D2H pageable in 5 copies (H1,H2,prob,match,dosage)", best * 1e3);
	}
	return 0;
}
