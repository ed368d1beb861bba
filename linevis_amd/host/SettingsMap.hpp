// SettingsMap.hpp -- string->string settings with the accessors of the reference's SettingsMap
// (src/Utils/InternalState.hpp:43-125): same keys, same encodings ("true"/"1" for bools, decimal numbers).
#pragma once

#include <cstdlib>
#include <map>
#include <sstream>
#include <string>

namespace lv {

class SettingsMap {
public:
    SettingsMap() = default;
    explicit SettingsMap(const std::map<std::string, std::string>& stringMap) : settings(stringMap) {}

    std::string getValue(const char* key) const {
        auto it = settings.find(key);
        return it == settings.end() ? "" : it->second;
    }
    int getIntValue(const char* key) const { return std::atoi(getValue(key).c_str()); }
    float getFloatValue(const char* key) const { return std::strtof(getValue(key).c_str(), nullptr); }
    bool getBoolValue(const char* key) const {
        std::string val = getValue(key);
        if (val == "false" || val == "0") return false;
        return val.length() > 0;
    }
    void addKeyValue(const std::string& key, const std::string& value) { settings[key] = value; }
    void addKeyValue(const std::string& key, const char* value) { settings[key] = value; }
    void addKeyValue(const std::string& key, bool value) { settings[key] = value ? "true" : "false"; }
    template <typename T>
    void addKeyValue(const std::string& key, const T& value) {
        std::ostringstream os;
        os.precision(9);
        os << value;
        settings[key] = os.str();
    }
    bool isEmpty() const { return settings.empty(); }
    void clear() { settings.clear(); }

    bool getValueOpt(const char* key, std::string& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = it->second;
        return true;
    }
    bool getValueOpt(const char* key, bool& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = (it->second == "true") || (it->second == "1");
        return true;
    }
    bool getValueOpt(const char* key, float& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = std::strtof(it->second.c_str(), nullptr);
        return true;
    }
    bool getValueOpt(const char* key, int& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = int(std::strtol(it->second.c_str(), nullptr, 10));
        return true;
    }
    bool getValueOpt(const char* key, uint32_t& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = uint32_t(std::strtoul(it->second.c_str(), nullptr, 10));
        return true;
    }
    const std::map<std::string, std::string>& getMap() const { return settings; }
    bool operator==(const SettingsMap& rhs) const { return settings == rhs.settings; }
    bool operator!=(const SettingsMap& rhs) const { return !(*this == rhs); }

private:
    std::map<std::string, std::string> settings;
};

} // namespace lv
