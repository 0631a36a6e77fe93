// STUB for tests/test_adapters.py: just enough of CoinUtils (absent from this tree) to type-check the
// adapters under include/adapters/.  Signatures follow their uses in the reference tree.
#ifndef CoinHelperFunctions_STUB
#define CoinHelperFunctions_STUB
#include <cstddef>
#include <cstring>
#define COIN_RESTRICT
typedef int CoinBigIndex;
template < class T > inline void CoinMemcpyN(const T *from, int size, T *to) { std::memcpy(to, from, sizeof(T) * (size_t)size); }
#endif
