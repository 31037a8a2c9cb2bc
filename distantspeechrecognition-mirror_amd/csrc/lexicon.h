// csrc/lexicon.h -- Lexicon (asr/dictionary/distribTree.h:40-65, distribTree.cc:36-87; List<String>: btk/common/mlist.h:71-250): symbol <-> index,
// indices are line order (the index column of the file is parsed and ignored), ';' starts a comment line, a repeated symbol is skipped.
#pragma once
#include "common.h"
#include <string>
#include <unordered_map>
#include <vector>

struct dsr_lexicon {
  std::string name; std::vector<std::string> syms; std::unordered_map<std::string, unsigned> idx;
  void clear() { syms.clear(); idx.clear(); }
  bool isPresent(const std::string& s) const { return idx.find(s) != idx.end(); }
  unsigned add(const std::string& s) { const unsigned i = (unsigned) syms.size(); idx.emplace(s, i); syms.push_back(s); return i; }
  unsigned index(const std::string& s, bool create = false) {
    auto it = idx.find(s);
    if (it != idx.end()) return it->second;
    if (!create) throw dsr::Error(DSR_E_KEY, "Could not find key %s in list %s", s.c_str(), name.c_str());        // List::index -> jkey_error (mlist.h:109-114)
    return add(s);
  }
  const std::string& symbol(unsigned i) const {
    if (i >= syms.size()) throw dsr::Error(DSR_E_INDEX, "Index %u out of range (%zu) in list %s", i, syms.size(), name.c_str());
    return syms[i];
  }
  void read(const char* fileName);
  void write(const char* fileName, bool writeHeader) const;
};
