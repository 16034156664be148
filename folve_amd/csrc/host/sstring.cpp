#include "sstring.h"

namespace folve {

namespace {
inline bool is_ctl(unsigned char c) { return c < 0x20 || c == 0x7f; }
}  // namespace

int sstring(const char* srce, char* dest, int size) {
    if (size < 0) return 0;
    const char kSingle = '\'', kDouble = '"', kEscape = '\\';
    char open_quote = 0;
    bool escape_next = false;
    int rd = 0, wr = 0;
    while (wr != size) {
        unsigned char c = static_cast<unsigned char>(srce[rd++]);
        if (c == '\t') c = ' ';
        if (is_ctl(c)) {
            // end of input: fine for a bare token, an error inside quotes or after a backslash
            if (open_quote || escape_next) break;
            dest[wr] = 0;
            return rd - 1;
        }
        if (escape_next) {
            dest[wr++] = static_cast<char>(c);
            escape_next = false;
        } else if (c == kEscape) {
            if (open_quote == kSingle) dest[wr++] = static_cast<char>(c);   // no escapes in '...'
            else escape_next = true;
        } else if (c == kSingle || c == kDouble) {
            if (c == static_cast<unsigned char>(open_quote)) {              // closing quote
                dest[wr] = 0;
                return rd;
            }
            if (open_quote || wr) break;                                    // stray quote
            open_quote = static_cast<char>(c);
        } else if (c == ' ') {
            if (open_quote) dest[wr++] = ' ';
            else if (wr) {                                                  // token ends
                dest[wr] = 0;
                return rd - 1;
            }                                                               // else: leading blank
        } else {
            dest[wr++] = static_cast<char>(c);
        }
    }
    dest[0] = 0;     // output full, or a malformed token
    return 0;
}

}  // namespace folve
