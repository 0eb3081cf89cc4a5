// C++ test of the host mirror (paillier_amd/host/paillier.hpp) against golden vectors exported by the pytest driver
// (tests/test_gpu_cpp_host.py writes them from tests/golden/*.json as "key value-hex ..." lines).  Mirrors the shape of the
// reference's tests: TestEncryptDecrypt, TestAddCiphertext, TestSubCiphertext, TestMulCiphertext, TestDecryption.
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "../../paillier_amd/host/paillier.hpp"

using namespace paillier;

static std::map<std::string, std::vector<Int>> load(const char* path) {
  std::map<std::string, std::vector<Int>> m;
  std::ifstream f(path);
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string key, tok;
    ss >> key;
    while (ss >> tok) m[key].push_back(Int::FromHex(tok));
  }
  return m;
}

#define EXPECT(cond)                                                                    \
  do {                                                                                  \
    if (!(cond)) { std::cerr << "FAILED: " #cond " (" << __FILE__ << ":" << __LINE__ << ")\n"; return 1; } \
  } while (0)

static std::vector<Int> values(const std::vector<Ciphertext>& cts) {
  std::vector<Int> v;
  for (auto& c : cts) v.push_back(c.C);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "usage: test_host_mirror vectors.txt\n"; return 2; }
  auto v = load(argv[1]);
  GPU gpu(0);
  PublicKey pk(gpu, v["n"][0], v["g"][0]);
  SecretKey sk(gpu, pk, v["lambda"][0]);

  // TestEncryptDecrypt (paillier_test.go:52-63) on golden (m, r, c)
  auto cts = pk.EncryptWithRBatch(v["enc_m"], v["enc_r"]);
  EXPECT(values(cts) == v["enc_c"]);
  EXPECT(sk.DecryptBatch(cts) == v["enc_m"]);

  // golden Decrypt vectors include non-units and zero
  std::vector<Ciphertext> dc;
  for (auto& c : v["dec_c"]) dc.push_back({c, EncLevelOne});
  EXPECT(sk.DecryptBatch(dc) == v["dec_m"]);

  // TestAddCiphertext / TestMulCiphertext (operations_test.go:11-50,72-90)
  std::vector<Ciphertext> a, b;
  for (auto& c : v["add_a"]) a.push_back({c, EncLevelOne});
  for (auto& c : v["add_b"]) b.push_back({c, EncLevelOne});
  EXPECT(values(pk.AddBatch(a, b)) == v["add_out"]);
  EXPECT(values(pk.ConstMultBatch(a, v["cm_k0"][0])) == v["cm_out"]);
  // TestSubCiphertext: Sub(Add(a, b), b) == a for unit ciphertexts
  auto ab = pk.AddBatch(cts, cts);
  EXPECT(values(pk.SubBatch(ab, cts)) == values(cts));

  // threshold: TestDecryption toy KAT (thresholdkey_test.go:267-281) through the C++ mirror
  ThresholdPublicKey tk(gpu, Int(637753), Int(637754), 2, 2);
  auto m = tk.CombinePartialDecryptionsBatch({{1, {Int(384111638639ull)}}, {2, {Int(235243761043ull)}}});
  EXPECT(m.size() == 1 && m[0] == Int(100));
  try {
    tk.CombinePartialDecryptionsBatch({{1, {Int(384111638639ull)}}});
    EXPECT(false);
  } catch (const Error& e) {
    EXPECT(e.code == PGPU_ERR_THRESHOLD);
  }
  std::cout << "host mirror ok: " << cts.size() << " encryptions, " << dc.size() << " decryptions\n";
  return 0;
}
