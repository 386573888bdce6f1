#include "utilities.h"

#include <sstream>

namespace utilityCore {

lin::mat4 buildTransformationMatrix(lin::vec3 translation, lin::vec3 rotation, lin::vec3 scale) {
    using namespace lin;
    const mat4 T = translate(mat4(), translation);
    mat4 R = rotate(mat4(), rotation.x * (float)PI / 180, vec3(1, 0, 0));
    R = R * rotate(mat4(), rotation.y * (float)PI / 180, vec3(0, 1, 0));
    R = R * rotate(mat4(), rotation.z * (float)PI / 180, vec3(0, 0, 1));
    const mat4 S = lin::scale(mat4(), scale);
    return T * R * S;
}

std::vector<std::string> tokenizeString(const std::string &str) {
    std::vector<std::string> out;
    std::istringstream in(str);
    for (std::string tok; in >> tok;) out.push_back(tok);
    return out;
}

std::istream &safeGetline(std::istream &is, std::string &t) {
    t.clear();
    std::istream::sentry guard(is, true);
    std::streambuf *buf = is.rdbuf();
    while (true) {
        const int ch = buf->sbumpc();
        if (ch == '\n') break;
        if (ch == '\r') {
            if (buf->sgetc() == '\n') buf->sbumpc();
            break;
        }
        if (ch == std::streambuf::traits_type::eof()) {
            if (t.empty()) is.setstate(std::ios::eofbit);  // last line without a line ending is still a line
            break;
        }
        t.push_back((char)ch);
    }
    return is;
}

}  // namespace utilityCore
