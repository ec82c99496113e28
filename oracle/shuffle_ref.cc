// ORACLE (test infrastructure) -- the real libstdc++ std::random_shuffle over the real libc rand(),
// exactly what operator_cxx/proposal_target.cc:83,102,118 calls.  Used to pin the restated
// shuffle + glibc TYPE_3 generator of oracle/proposal_target.c.  Built with -std=c++11
// (random_shuffle was removed in C++17).
#include <algorithm>
#include <cstdlib>
#include <vector>

extern "C" void orc_std_random_shuffle(unsigned* a, int n) {
  std::vector<unsigned> v(a, a + n);
  std::random_shuffle(v.begin(), v.end());
  std::copy(v.begin(), v.end(), a);
}
extern "C" void orc_libc_srand(unsigned seed) { std::srand(seed); }
extern "C" int orc_libc_rand(void) { return std::rand(); }
