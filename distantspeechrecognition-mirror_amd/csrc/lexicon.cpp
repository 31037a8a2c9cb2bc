// csrc/lexicon.cpp -- see lexicon.h.  Host-side container of the boundary (no device work).
#include "lexicon.h"
#include <ctime>

void dsr_lexicon::read(const char* fileName)
{
  if (!fileName || !*fileName) throw dsr::Error(DSR_E_IO, "File name is null.");
  clear();
  FILE* fp = fopen(fileName, "r");
  if (!fp) throw dsr::Error(DSR_E_IO, "Could not open file %s", fileName);
  char* line = nullptr; size_t cap = 0;
  while (getline(&line, &cap, fp) > 0) {
    if (line[0] == ';') continue;                              // _commentChar (distribTree.cc:39,62)
    char* t0 = strtok(line, " \t\n");
    if (!t0) continue;                                         // (the reference builds a String from NULL on a blank line)
    (void) strtok(nullptr, " \t\n");                           // the index column is read and never used (:69-71,83)
    const std::string s(t0);
    if (isPresent(s)) continue;                                // "Symbol %s already exists." (:75-79)
    add(s);
  }
  free(line); fclose(fp);
}

void dsr_lexicon::write(const char* fileName, bool writeHeader) const
{
  // Lexicon::write (distribTree.cc:89-122): optional ';' header (name, type, item count, date), then "%30s %10d\n" per symbol in index order
  if (!fileName || !*fileName) throw dsr::Error(DSR_E_IO, "File name is null.\n");
  if (syms.empty()) throw dsr::Error(DSR_E_IO, "Lexicon '%s' has no entries.\n", name.c_str());
  FILE* fp = fopen(fileName, "w");
  if (!fp) throw dsr::Error(DSR_E_IO, "Could not open file %s", fileName);
  if (writeHeader) {
    time_t t = time(nullptr);
    fprintf(fp, "; -------------------------------------------------------\n");
    fprintf(fp, ";  Name            : %s\n", name.c_str());
    fprintf(fp, ";  Type            : Lexicon\n");
    fprintf(fp, ";  Number of Items : %d\n", (int) syms.size());
    fprintf(fp, ";  Date            : %s", ctime(&t));
    fprintf(fp, "; -------------------------------------------------------\n");
  }
  for (size_t i = 0; i < syms.size(); i++) fprintf(fp, "%30s %10d\n", syms[i].c_str(), (int) i);
  fclose(fp);
}

using namespace dsr;
extern "C" {

dsr_status dsr_lexicon_create(const char* name, const char* fileName, dsr_lexicon** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    dsr_lexicon* l = new dsr_lexicon(); l->name = name ? name : "Lexicon";
    try { if (fileName && *fileName) l->read(fileName); } catch (...) { delete l; throw; }
    *out = l;
  });
}
void dsr_lexicon_destroy(dsr_lexicon* l) { delete l; }
dsr_status dsr_lexicon_read(dsr_lexicon* l, const char* fileName) { return guard([&] { if (!l) throw Error(DSR_E_PARAMETER, "null argument"); l->read(fileName); }); }
dsr_status dsr_lexicon_write(const dsr_lexicon* l, const char* fileName, int writeHeader) { return guard([&] { if (!l) throw Error(DSR_E_PARAMETER, "null argument"); l->write(fileName, writeHeader != 0); }); }
dsr_status dsr_lexicon_clear(dsr_lexicon* l) { return guard([&] { if (!l) throw Error(DSR_E_PARAMETER, "null argument"); l->clear(); }); }
int dsr_lexicon_size(const dsr_lexicon* l) { return l ? (int) l->syms.size() : 0; }
const char* dsr_lexicon_name(const dsr_lexicon* l) { return l ? l->name.c_str() : ""; }
int dsr_lexicon_is_present(const dsr_lexicon* l, const char* symbol) { return (l && symbol && l->isPresent(symbol)) ? 1 : 0; }
dsr_status dsr_lexicon_index(dsr_lexicon* l, const char* symbol, int create, unsigned* index)
{ return guard([&] { if (!l || !symbol || !index) throw Error(DSR_E_PARAMETER, "null argument"); *index = l->index(symbol, create != 0); }); }
dsr_status dsr_lexicon_symbol(const dsr_lexicon* l, unsigned index, const char** symbol)
{ return guard([&] { if (!l || !symbol) throw Error(DSR_E_PARAMETER, "null argument"); *symbol = l->symbol(index).c_str(); }); }

}  // extern "C"
